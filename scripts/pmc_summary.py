"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, the mean of every counter over the dispatches and a few
derived ratios (only between counters of the same unit, or against the dispatch's wall time).
usage: pmc_summary.py counter_collection.csv [skip_first_n_dispatches_per_kernel]"""
import csv
import sys
from collections import defaultdict

skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
seen = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        seen[k][c] += 1
        if seen[k][c] > skip:
            acc[k][c].append(float(r["Counter_Value"]))
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = sorted({c for k in acc for c in acc[k]})
print("%-70s %6s " % ("kernel", "calls") + " ".join("%24s" % c[-24:] for c in names))
for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
    n = max(len(v) for v in acc[k].values())
    means = {c: (sum(acc[k][c]) / len(acc[k][c]) if acc[k][c] else float("nan")) for c in names}
    print("%-70s %6d " % (k[:70], n) + " ".join("%24.4g" % means[c] for c in names))
    d = []
    ns = sum(dur[k]) / len(dur[k])
    d.append("%.1f us under the profiler" % (ns / 1e3))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in means:
        # cycles summed over SIMDs (checked: conv2_fwd issues 70 000 MFMA 32x32x16 and counts 2.24e6 = 32 cycles each)
        # -> share of all 1024 SIMDs' cycles, at 2.4 GHz nominal, with the MFMA pipe busy
        d.append("MFMA pipe busy on %.1f%% of the chip's SIMD-cycles" % (100 * means["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * 2.4 * 1024)))
    if means.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in means:
        d.append("LDS bank-conflict cycles / LDS active = %.1f%%" % (100 * means["SQ_LDS_BANK_CONFLICT"] / means["SQ_LDS_IDX_ACTIVE"]))
    if means.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in means:
        d.append("waiting on any instruction / wave cycles = %.1f%%" % (100 * means["SQ_WAIT_INST_ANY"] / means["SQ_WAVE_CYCLES"]))
    if d:
        print("      " + "; ".join(d))
