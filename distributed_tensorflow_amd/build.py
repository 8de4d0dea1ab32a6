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
import re
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
    # location independent: the tree is copied to another path on the GPU box and must hash the same there
    # (the compiler is named by its base name: the same toolchain can sit behind /opt/rocm or /opt/rocm-x.y.z on PATH)
    extra = " ".join(os.path.basename(w) if w in (HIPCC, CXX) else w for w in extra.split(" "))
    h = hashlib.sha1(extra.replace(ROOT, "$ROOT").encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(os.path.relpath(p, ROOT).encode())
            h.update(f.read())
    return h.hexdigest()


def _run(cmd: List[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build command failed:\n  %s\n%s" % (" ".join(cmd), r.stdout))


def _headers(d: str) -> List[str]:
    return [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".h")]


_MARK = b"DTF_SOURCE_DIGEST="


def embedded_digest(lib_path: str) -> str:
    """The source digest compiled INTO a library (read from the file's bytes: nothing is loaded)."""
    try:
        with open(lib_path, "rb") as f:
            m = re.search(_MARK + b"([0-9a-f]{40})", f.read())
    except OSError:
        return ""
    return m.group(1).decode() if m else ""


def _sources(srcdir: str, exts) -> List[str]:
    return sorted(os.path.join(srcdir, f) for f in os.listdir(srcdir) if f.endswith(exts))


def source_digest(which: str) -> str:
    """Digest of the sources + build line that ``libdtf_<which>.so`` must have been built from."""
    srcdir, exts, cc, ld, _ = _recipe(which)
    return _digest(_sources(srcdir, exts) + _headers(srcdir), " ".join(cc + ld))


def is_current(which: str) -> bool:
    return embedded_digest(_recipe(which)[4]) == source_digest(which)


def _build_lib(which: str, force: bool, jobs: int) -> str:
    srcdir, exts, compiler_cmd, link_cmd, out = _recipe(which)
    srcs = _sources(srcdir, exts)
    if not srcs:
        raise RuntimeError("no sources in " + srcdir)
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    digest = _digest(srcs + _headers(srcdir), " ".join(compiler_cmd + link_cmd))
    if not force and os.path.exists(out) and embedded_digest(out) == digest:
        return out
    # the digest travels inside the library (dtf_<which>_source_digest()), not in a side-car file that can go stale
    gen = os.path.join(OBJDIR, "dtf_%s_digest.cc" % which)
    with open(gen, "w") as f:
        f.write('static const char kDigest[] = "%s%s";\n'
                'extern "C" __attribute__((visibility("default"), used)) const char* dtf_%s_source_digest() '
                '{ return kDigest + %d; }\n' % (_MARK.decode(), digest, which, len(_MARK)))
    pairs = [(s, os.path.join(OBJDIR, os.path.basename(s) + ".o")) for s in srcs]
    pairs.append((gen, gen + ".o"))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda so: _run(compiler_cmd + (["-x", "c++"] if which == "hip" and so[0] == gen else []) +
                                    ["-c", so[0], "-o", so[1]]), pairs))
    _run(link_cmd + [o for _, o in pairs] + ["-o", out])
    if embedded_digest(out) != digest:
        raise RuntimeError("%s does not carry its source digest" % out)
    return out


def _recipe(which: str):
    if which == "hip":
        d = os.path.join(CSRC, "hip")
        cc = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-I", d]
        cc += os.environ.get("DTF_EXTRA_HIPCC_FLAGS", "").split()  # (experiments; part of the digest)
        ld = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC"]
        return d, (".hip",), cc, ld, os.path.join(LIBDIR, "libdtf_hip.so")
    d = os.path.join(CSRC, "host")
    cc = [CXX, "-O2", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-pthread", "-I", d]
    ld = [CXX, "-shared", "-fPIC", "-pthread"]
    return d, (".cc",), cc, ld, os.path.join(LIBDIR, "libdtf_host.so")


def build_hip(force: bool = False, jobs: int = 4) -> str:
    return _build_lib("hip", force, jobs)


def build_host(force: bool = False, jobs: int = 4) -> str:
    return _build_lib("host", force, jobs)


SANITIZERS = {
    # CPU only (the GPU side has no sanitizer on this pool).  The runtimes are linked statically into a native
    # self-test executable, so nothing has to be injected into the Python interpreter.
    "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-static-libubsan"],
    "tsan": ["-fsanitize=thread", "-static-libtsan"],
}


def build_host_selftest(sanitizer: str = "", outdir: str = "") -> str:
    """``csrc/host/*.cc`` + ``csrc/host/selftest/host_selftest.cc`` as one executable, optionally under a sanitizer."""
    d = os.path.join(CSRC, "host")
    outdir = outdir or os.path.join(ROOT, "build", "san")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "host_selftest" + ("_" + sanitizer if sanitizer else ""))
    flags = SANITIZERS[sanitizer] if sanitizer else []
    _run([CXX, "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-pthread", "-I", d] + flags +
         _sources(d, (".cc",)) + [os.path.join(d, "selftest", "host_selftest.cc"), "-o", out])
    return out


def build_all(force: bool = False) -> List[str]:
    return [build_host(force), build_hip(force)]


if __name__ == "__main__":
    outs = build_all(force="--force" in sys.argv)
    for o in outs:
        print("built", o)
