"""Per-layer table of the fine-tune conv passes at one batch size: geometry, GEMM shape, chosen core, time and TFLOP/s
for forward / data gradient / filter gradient (from the trainer's own autotune timings).
    python scripts/inception_layer_table.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.models.inception_train import InceptionTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tr = InceptionTrainer(5, B, "cuda", lr=0.01, optimizer="sgd", seed=1)
print(tr.tuning_summary())
rows = []
for op in tr.graph.ops:
    if op.kind != "conv":
        continue
    s = op.conv
    ho, wo, _ = tr.graph.shapes[op.dst]
    fl = 2.0 * B * ho * wo * s.cout * s.kh * s.kw * s.cin
    t = tr.tuning[s.name]
    rec = {"name": s.name, "geom": "%dx%d %d->%d k%dx%d s%d" % (ho, wo, s.cin, s.cout, s.kh, s.kw, s.stride), "M": B * ho * wo,
           "N": s.cout, "K": s.kh * s.kw * s.cin, "fl": fl}
    for k in ("fwd", "bwd", "dw"):
        if k in t:
            c = tr.core[s.name][k]
            rec[k] = (c, t[k][c] * 1e3)
    rows.append(rec)
tot = {k: sum(r[k][1] for r in rows if k in r) for k in ("fwd", "bwd", "dw")}
print("totals (us): fwd %.0f bwd %.0f dw %.0f" % (tot["fwd"], tot["bwd"], tot["dw"]))
print("%-44s %-26s %7s %5s %6s | %s" % ("layer", "geometry", "M", "N", "K", "pass: core us TFLOP/s"))
for r in sorted(rows, key=lambda r: -sum(r[k][1] for k in ("fwd", "bwd", "dw") if k in r)):
    cells = ["%s c%d %6.1f %5.0f" % (k, r[k][0], r[k][1], r["fl"] / r[k][1] / 1e6) for k in ("fwd", "bwd", "dw") if k in r]
    print("%-44s %-26s %7d %5d %6d | %s" % (r["name"][-44:], r["geom"], r["M"], r["N"], r["K"], "  ".join(cells)))
