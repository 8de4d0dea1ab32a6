"""In-tree native build (no JIT cache): ``python -m distributed_tensorflow_amd.build``.

* ``csrc/hip/*.hip``  -> ``distributed_tensorflow_amd/_lib/libdtf_hip.so``  (hipcc, ``--offload-arch=gfx950`` only)
* ``csrc/host/*.cc``  -> ``distributed_tensorflow_amd/_lib/libdtf_host.so`` (g++; tfevents writer, CRC32C,
  TCP rendezvous store, checkpoint container, IDX/bottleneck text codecs)

hipcc cross-compiles without a GPU, so the build also runs on the CPU-only CI box; the resulting
``.so`` files are git-ignored but travel with the tree to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from typing import List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "distributed_tensorflow_amd", "_lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"


def _digest(paths: List[str], extra: str = "") -> str:
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build command failed:\n  %s\n%s" % (" ".join(cmd), r.stdout))


def _headers(d: str) -> List[str]:
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]


def _build_lib(srcdir: str, exts, compiler_cmd, link_cmd, out: str, force: bool, jobs: int) -> str:
    srcs = sorted(os.path.join(srcdir, f) for f in os.listdir(srcdir) if f.endswith(exts))
    if not srcs:
        raise RuntimeError("no sources in " + srcdir)
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = out + ".sha1"
    digest = _digest(srcs + _headers(srcdir), " ".join(compiler_cmd + link_cmd))
    if (not force and os.path.exists(out) and os.path.exists(stamp)
            and open(stamp).read().strip() == digest):
        return out
    objs = [os.path.join(OBJDIR, os.path.basename(s) + ".o") for s in srcs]
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda so: _run(compiler_cmd + ["-c", so[0], "-o", so[1]]), zip(srcs, objs)))
    _run(link_cmd + objs + ["-o", out])
    with open(stamp, "w") as f:
        f.write(digest)
    return out


def build_hip(force: bool = False, jobs: int = 4) -> str:
    cc = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
          "-I", os.path.join(CSRC, "hip")]
    ld = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"]
    return _build_lib(os.path.join(CSRC, "hip"), (".hip",), cc, ld, os.path.join(LIBDIR, "libdtf_hip.so"), force, jobs)


def build_host(force: bool = False, jobs: int = 4) -> str:
    cc = [CXX, "-O2", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-pthread", "-I", os.path.join(CSRC, "host")]
    ld = [CXX, "-shared", "-fPIC", "-pthread"]
    return _build_lib(os.path.join(CSRC, "host"), (".cc",), cc, ld, os.path.join(LIBDIR, "libdtf_host.so"), force, jobs)


def build_all(force: bool = False) -> List[str]:
    return [build_host(force), build_hip(force)]


if __name__ == "__main__":
    outs = build_all(force="--force" in sys.argv)
    for o in outs:
        print("built", o)
