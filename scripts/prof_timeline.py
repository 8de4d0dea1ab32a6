"""Print the kernel timeline of the last few steps from a rocprofv3 sqlite result: start / end relative to the first
kernel shown, duration, gap to the latest end so far, queue.  usage: prof_timeline.py results.db [n_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "name, start, end" + (", %s" % qcol if qcol else "")
rows = list(db.execute("select %s from kernels order by start desc limit %d" % (sel, n)))[::-1]
t0 = rows[0][1]
latest = rows[0][1]
print("%9s %9s %8s %8s %5s  %s" % ("start_us", "end_us", "dur_us", "gap_us", "queue", "kernel"))
for r in rows:
    gap = (r[1] - latest) / 1e3
    print("%9.2f %9.2f %8.2f %8.2f %5s  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap,
                                                  r[3] if qcol else "-", r[0][:90]))
    latest = max(latest, r[2])
