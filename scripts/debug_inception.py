import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.models import inception_v3 as iv3
from distributed_tensorflow_amd.models.retrain_head import RetrainHead
net = iv3.InceptionV3("cuda", seed=0)
for B in (8, 32, 64):
    x = torch.zeros(B, 299, 299, 8, device="cuda", dtype=torch.bfloat16)
    try:
        y = net.forward(x, use_graph=False)
        torch.cuda.synchronize()
        print("B", B, "ok", float(y.abs().mean()))
    except Exception as e:
        print("B", B, "FAILED", e)
torch.manual_seed(2)
N, C, B = 300, 5, 100
cache = torch.rand(N, 2048, device="cuda"); labels = torch.randint(0, C, (N,), device="cuda", dtype=torch.int32)
rows = torch.randint(0, N, (B,), device="cuda", dtype=torch.int32)
for ds in (True, False):
    hg = RetrainHead(C, "cuda", learning_rate=0.5, double_softmax=ds, seed=1); hc = RetrainHead(C, "cpu", learning_rate=0.5, double_softmax=ds, seed=1)
    hg.weights.copy_(torch.randn(2048, C, device="cuda") * 0.01); hc.P.copy_(hg.P.cpu())
    for i in range(3):
        hg.train_step(cache, labels, rows); hc.train_step(cache.cpu(), labels.cpu(), rows.cpu())
        print(ds, i, hg.last_metrics(), hc.last_metrics(), float((hg.P.cpu()-hc.P).abs().max()), float(hc.P.abs().max()))
