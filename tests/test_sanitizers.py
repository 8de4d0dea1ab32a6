"""Sanitizer builds of the host runtime (VERDICT r1 'Next round' 9): ``csrc/host`` + its native self-test
(``csrc/host/selftest/host_selftest.cc``, the C++ mirror of ``test_host_native.py``: CRC32C, events writer with a
concurrent flusher, rendezvous store under 8 client threads with connection churn, checkpoint container) are compiled
with ``-fsanitize=address,undefined`` and with ``-fsanitize=thread`` and run.  CPU only; the sanitizer runtimes are linked
statically into the executable so nothing is injected into the Python interpreter."""
import subprocess

import pytest

from distributed_tensorflow_amd import build


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_runtime_is_clean_under_sanitizer(san, tmp_path):
    exe = build.build_host_selftest(san, outdir=str(tmp_path))
    r = subprocess.run([exe, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if "unexpected memory mapping" in r.stdout:   # the ThreadSanitizer runtime vs. this kernel's address-space layout
        pytest.skip("sanitizer runtime cannot start on this kernel: " + r.stdout[-200:])
    assert r.returncode == 0 and "host selftest ok" in r.stdout, r.stdout[-4000:]
    for needle in ("ERROR: AddressSanitizer", "WARNING: ThreadSanitizer", "runtime error:", "LeakSanitizer"):
        assert needle not in r.stdout, r.stdout[-4000:]


def test_libraries_carry_the_digest_of_their_sources():
    """``build()`` no longer trusts a side-car stamp: the digest is compiled into the library."""
    from distributed_tensorflow_amd import native
    build.build_host()
    assert build.is_current("host") and len(build.embedded_digest(build._recipe("host")[4])) == 40
    assert native.host().dtf_host_source_digest().decode() == build.source_digest("host")
