"""Summarise a rocprofv3 sqlite result (kernel name, calls, total ms, %, avg us)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print("%-110s %7s %10s %6s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us"))
for r in rows[:n]:
    print("%-110s %7d %10.2f %6.1f %10.1f" % (r[0][:110], r[1], r[2] / 1e6, 100 * r[2] / tot, r[3] / 1e3))
print("total %.2f ms" % (tot / 1e6))
