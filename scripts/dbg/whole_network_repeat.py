"""Repeat the whole-network layerwise check N times in one process (a fresh trainer and a fresh autotune each time) and
print each run's margins; stops at the first failure and prints it.  Usage: python scripts/dbg/whole_network_repeat.py [N]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_whole_network_gpu as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(n):
    t0 = time.time()
    try:
        t.test_full_network_step_matches_autograd_layerwise()
    except AssertionError:
        traceback.print_exc()
        print("run %d FAILED" % i, flush=True)
        sys.exit(1)
    print("run %d ok in %.1f s" % (i, time.time() - t0), flush=True)
