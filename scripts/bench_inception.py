#!/usr/bin/env python
"""Bottleneck-extraction throughput of the batched Inception-v3 forward (images/sec, 1 GPU)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from distributed_tensorflow_amd.models import inception_v3 as iv3  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
use_graph = (sys.argv[3] != "nograph") if len(sys.argv) > 3 else True
core = sys.argv[4] if len(sys.argv) > 4 else "auto"
net = iv3.InceptionV3("cuda", seed=0, conv_core=core)
x = torch.zeros(B, 299, 299, 8, device="cuda", dtype=torch.bfloat16)
x[..., :3] = (torch.rand(B, 299, 299, 3, device="cuda") * 2 - 1).to(torch.bfloat16)
for _ in range(2):
    net.forward(x, use_graph=use_graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    net.forward(x, use_graph=use_graph)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = iv3.graph_flops(net.graph) * B
print(json.dumps({"metric": "inception_v3 bottleneck extraction", "batch": B, "ms_per_batch": round(dt * 1e3, 3),
                  "images_per_sec": round(B / dt, 1), "tflops": round(fl / dt / 1e12, 2), "hipgraph": use_graph,
                  "conv_core": core, "v2_layers": sum(1 for v in net._core.get(B, {}).values() if v >= 2)}))
