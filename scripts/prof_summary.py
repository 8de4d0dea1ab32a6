"""Summarise a rocprofv3 sqlite result (kernel name, calls, total ms, %, avg us).

usage: prof_summary.py results.db [top_n] [tail_fraction]
``tail_fraction`` (0..1) keeps only kernels that started in the last fraction of the traced interval, which
drops start-up work (autotuning, weight packing, BN calibration) from a steady-state step profile.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tail = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
cut = t1 - tail * (t1 - t0)
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where start >= ? "
                        "group by name order by 3 desc", (cut,)))
tot = sum(r[2] for r in rows)
print("%-110s %7s %10s %6s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us"))
for r in rows[:n]:
    print("%-110s %7d %10.2f %6.1f %10.1f" % (r[0][:110], r[1], r[2] / 1e6, 100 * r[2] / tot, r[3] / 1e3))
print("total %.2f ms of kernel time in a %.2f ms window" % (tot / 1e6, (t1 - cut) / 1e6))
