"""Duration histogram of the kernels whose name contains a substring, from a rocprofv3 sqlite result.

usage: prof_hist.py results.db substring [tail_fraction]
Prints, per duration bucket (us), the number of calls and their share of the kernel's total time.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
sub = sys.argv[2]
tail = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
cut = t1 - tail * (t1 - t0)
d = [r[0] / 1e3 for r in cur.execute("select end-start from kernels where start >= ? and name like ?", (cut, "%" + sub + "%"))]
tot = sum(d)
edges = [0, 3, 5, 7, 10, 15, 20, 30, 50, 100, 200, 1e9]
print("%s: %d calls, %.2f ms" % (sub, len(d), tot / 1e3))
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = [x for x in d if lo <= x < hi]
    if sel:
        print("  %5.0f-%-5s us: %5d calls, %5.1f %% of its time, mean %.1f us" % (lo, "%.0f" % hi if hi < 1e8 else "inf", len(sel), 100 * sum(sel) / tot, sum(sel) / len(sel)))
