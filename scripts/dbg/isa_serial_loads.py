"""Diagnostic: find chains of serialised global loads in the gfx950 ISA of csrc/hip/*.hip.

A load under `if (in range)` — or a clamped load whose value is only used under such an `if` — is compiled to a branch that
holds the load AND its `s_waitcnt vmcnt(0)`; N of them in an unrolled loop are N dependent round trips to L2 (found in the
fine-tune epilogues, the pool kernels and the MNIST fp32 data-gradient prologue).  This script compiles every source to
assembly (device only) and reports, per kernel, runs of `s_waitcnt vmcnt(0)` that each cover exactly ONE load.

    python scripts/dbg/isa_serial_loads.py [min_run]      (CPU box: hipcc cross-compiles)
"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
min_run = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tmp = tempfile.mkdtemp()
procs = []
for src in sorted(glob.glob(os.path.join(ROOT, "csrc", "hip", "*.hip"))):
    out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
    procs.append((src, out, subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src,
                                              "-o", out, "-I", os.path.join(ROOT, "csrc", "hip")], stderr=subprocess.DEVNULL)))
load = re.compile(r"\b(global_load|buffer_load|flat_load|scratch_load)")
found = 0
for src, out, p in procs:
    p.wait()
    if not os.path.exists(out):
        print("could not compile", src)
        continue
    kern, since, run, best = None, 0, 0, {}
    for line in open(out):
        m = re.match(r"^(\w+):\s*;\s*@", line)
        if m:
            kern, since, run = m.group(1), 0, 0
        t = line.strip()
        if load.search(t) and " lds" not in t:
            since += 1
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            run = run + 1 if since == 1 else 0
            since = 0
            if kern and run > best.get(kern, 0):
                best[kern] = run
        elif t.startswith("s_waitcnt") and "vmcnt(" in t:
            run, since = 0, 0
    for k, r in sorted(best.items(), key=lambda kv: -kv[1]):
        if r >= min_run:
            found += 1
            print("%-18s %3d single-load waits in a row   %s" % (os.path.basename(src), r, k[:110]))
print("%d kernel(s) with a run of >= %d serialised loads" % (found, min_run))

# ---- second pass: loops whose body requests at most two loads before it waits for all of them (a streaming loop with one
# or two loads in flight per lane: the bandwidth-latency product of HBM needs more)
print()
print("loops with <= 2 loads in flight per iteration (body has a full vmcnt(0) wait):")
for src, out, p in procs:
    if not os.path.exists(out):
        continue
    lines = open(out).read().split("\n")
    kern, labels = None, {}
    for i, line in enumerate(lines):
        m = re.match(r"^(\w+):\s*;\s*@", line)
        if m:
            kern, labels = m.group(1), {}
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            labels[m.group(1)] = i
            continue
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", line)
        if m and m.group(1) in labels and kern:  # backward branch = loop
            body = lines[labels[m.group(1)]:i]
            if any(re.match(r"^\.LBB", b) for b in body[1:]):
                continue  # (only innermost, single-block loops)
            nload = sum(1 for b in body if load.search(b) and " lds" not in b)
            full = sum(1 for b in body if "s_waitcnt" in b and "vmcnt(0)" in b)
            nmfma = sum(1 for b in body if "v_mfma" in b)
            if 1 <= nload <= 2 and full >= 1 and nmfma == 0:
                print("%-18s %d load(s) per iteration   %s" % (os.path.basename(src), nload, kern[:100]))
