"""Run the native Inception forward+backward several times on the same trainer, inputs and cores and report, per tensor
family, the largest run-to-run relative L2 difference (fp32 atomics reorder sums; anything above ~1e-5 wants explaining)."""
import sys

import torch

sys.path.insert(0, ".")
from distributed_tensorflow_amd.models.inception_train import InceptionTrainer  # noqa: E402

BF = torch.bfloat16
torch.manual_seed(0)
B, C = 4, 5
tr = InceptionTrainer(C, B, "cuda", lr=0.01, seed=3)
x = torch.zeros(B, 299, 299, 8, device="cuda")
x[..., :3] = torch.randn(B, 299, 299, 3, device="cuda")
x = x.to(BF)
labels = torch.tensor([0, 3, 1, 4], device="cuda", dtype=torch.int32)


def l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


ref = None
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    tr.G.zero_()
    tr.fwd_bwd(x, labels)
    torch.cuda.synchronize()
    snap = {"G": tr.G.clone(), "act": {k: v.clone() for k, v in tr.act.items()}, "dact": {k: v.clone() for k, v in tr.dact.items()}}
    if ref is None:
        ref = snap
        continue
    worst = {}
    for fam in ("act", "dact"):
        w = max(((l2(snap[fam][k], ref[fam][k]), k) for k in ref[fam]), key=lambda t: t[0])
        worst[fam] = w
    per = []
    for op in tr.graph.ops:
        if op.kind == "conv":
            n = op.conv.name + "/conv2d_params"
            per.append((l2(tr.layout.view(snap["G"], n), tr.layout.view(ref["G"], n)), op.conv.name))
    worst["dW"] = max(per, key=lambda t: t[0])
    print("run %d vs run 0:" % run, {k: ("%.2e" % v[0], v[1]) for k, v in worst.items()}, flush=True)
    if run == 1:
        seen = set()
        for op in tr.graph.ops:
            if op.dst in seen or op.dst not in ref["act"]:
                continue
            seen.add(op.dst)
            a, b = snap["act"][op.dst].float(), ref["act"][op.dst].float()
            print("  act %-28s rel L2 %.3e  elements differing %d of %d" % (op.dst, l2(a, b), int((a != b).sum()), a.numel()), flush=True)
