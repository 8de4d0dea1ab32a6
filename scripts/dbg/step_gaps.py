"""Mean gap in front of every kernel of the training loop, by kernel name, and the gap in front of the step's first kernel
by position in the graph launch.  usage: step_gaps.py results.db [steps_per_graph]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
spg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(db.execute("select name, start, end from kernels order by start"))
first = [i for i, r in enumerate(rows) if r[0].startswith("void conv1_fwd_kernel")]
# the training loop: the longest run of conv1 launches spaced less than 400 us apart
best, cur = (0, 0), [0, 0]
for a in range(1, len(first)):
    if rows[first[a]][1] - rows[first[a - 1]][1] < 400e3:
        cur[1] = a
    else:
        cur = [a, a]
    if cur[1] - cur[0] > best[1] - best[0]:
        best = (cur[0], cur[1])
lo, hi = first[best[0]], first[best[1]]
gaps, durs = defaultdict(list), defaultdict(list)
latest = rows[lo][2]
for r in rows[lo + 1:hi]:
    gaps[r[0][:60]].append((r[1] - latest) / 1e3)
    durs[r[0][:60]].append((r[2] - r[1]) / 1e3)
    latest = max(latest, r[2])
nsteps = best[1] - best[0]
print("%d steps, period %.2f us" % (nsteps, (rows[hi][1] - rows[lo][1]) / 1e3 / nsteps))
print("%-62s %7s %9s %9s" % ("kernel", "calls", "dur_us", "gap_us"))
for k in gaps:
    print("%-62s %7d %9.2f %9.2f" % (k, len(gaps[k]), sum(durs[k]) / len(durs[k]), sum(gaps[k]) / len(gaps[k])))
pos = defaultdict(list)
prev_end = None
for j in range(best[0] + 1, best[1]):
    i = first[j]
    pos[(j - best[0]) % spg].append((rows[i][1] - max(r[2] for r in rows[max(lo, i - 3):i])) / 1e3)
print("gap before conv1 by (step index mod %d): " % spg + " ".join("%d:%.1f" % (p, sum(v) / len(v)) for p, v in sorted(pos.items())))
