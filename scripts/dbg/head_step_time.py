"""Diagnostic: time of one retrain-head SGD step (2048 -> C softmax head on the HBM bottleneck cache, batch 100)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from distributed_tensorflow_amd.models.retrain_head import RetrainHead
C, N, B = 5, 4000, 100
torch.manual_seed(0)
cache = torch.rand(N, 2048, device="cuda")
labels = torch.randint(0, C, (N,), device="cuda", dtype=torch.int32)
head = RetrainHead(C, "cuda", seed=1)
rows = torch.randint(0, N, (B,), device="cuda", dtype=torch.int32)
ref = RetrainHead(C, "cpu", seed=1)
for _ in range(3):
    head.train_step(cache, labels, rows)
    ref.train_step(cache.cpu(), labels.cpu(), rows.cpu())
print("max |W - cpu reference| after 3 steps: %.3e" % float((head.weights.cpu() - ref.weights).abs().max()))
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    head.train_step(cache, labels, rows)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%.1f us per head step, %.2f M bottleneck-images/s" % (1e6 * dt, B / dt / 1e6))
