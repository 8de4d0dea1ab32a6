"""Static check on the compiled gfx950 ISA (CPU only: hipcc cross-compiles).

A global load under `if (in range)` is compiled to a branch that holds the load AND its `s_waitcnt vmcnt(0)`; a run of them in
an unrolled loop is a chain of dependent round trips to L2.  Round 2 found such chains in the GEMM epilogues (bias, accumulate,
mask), the pool kernels and a conv prologue by reading the ISA (`scripts/dbg/isa_serial_loads.py`); this test keeps them out."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_kernel_waits_for_its_loads_one_by_one():
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "dbg", "isa_serial_loads.py"), "9"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "could not compile" not in r.stdout, r.stdout[-2000:]
    verdict = [l for l in r.stdout.splitlines() if "kernel(s) with a run of" in l]
    assert verdict and verdict[0].startswith("0 kernel(s)"), r.stdout[-3000:]
