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


def _kernel_bodies(src):
    """name -> ISA text of every kernel in one csrc/hip source, compiled for gfx950"""
    hip = os.path.join(REPO, "csrc", "hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", hip, "-S", "--cuda-device-only",
                        os.path.join(hip, src), "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name, buf = {}, None, []
    for line in r.stdout.splitlines():
        if line.startswith("_Z") and ":" in line.split(";")[0]:
            name, buf = line.split(":")[0], []
        elif name is not None:
            buf.append(line)
            if line.startswith(".Lfunc_end"):  # (not the first s_endpgm: a wave-specialised kernel has one per role)
                out[name] = "\n".join(buf)
                name = None
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_bf16x6_kernels_run_on_the_bf16_matrix_cores_without_scratch():
    """The fp32 path's bf16x6 kernels (csrc/hip/bf16x3.h): six v_mfma_f32_16x16x32_bf16 per product, no exact-fp32 MFMA left
    in them, nothing spilled; row-contiguous GEMM operands are read with the transposing LDS load."""
    conv = _kernel_bodies("f32x3_conv2.hip")
    for key in ("conv2_fwd_x3_kernel", "conv2_bwd_data_x3_kernel"):
        body = [b for n, b in conv.items() if key in n]
        assert len(body) == 1, (key, list(conv))
        assert body[0].count("v_mfma_f32_16x16x32_bf16") == 180, key  # 5 x 2 tiles x 6 plane products per k-unit; two units per loop pass + the odd one
        assert "v_mfma_f32_16x16x4_f32" not in body[0] and "scratch_" not in body[0], key
        assert "v_cvt_pk_bf16_f32" in body[0], key                     # the split happens where the operand is staged
    fc = _kernel_bodies("f32_fc.hip")
    fwd = [b for n, b in fc.items() if "fc_fwd32_kernelILb1ELb0" in n]
    assert len(fwd) == 1, list(fc)
    assert "v_mfma_f32_16x16x32_bf16" in fwd[0] and "ds_read_b64_tr_b16" in fwd[0] and "scratch_" not in fwd[0]
    exact = [b for n, b in fc.items() if "fc_fwd32_kernelILb0ELb0" in n]
    assert len(exact) == 1 and "v_mfma_f32_16x16x4_f32" in exact[0] and "v_mfma_f32_16x16x32_bf16" not in exact[0]
