"""Run the three conv GEMM passes of ONE Inception layer geometry in a loop (for rocprofv3 --pmc / --kernel-trace).

    python scripts/bench_conv_layer.py <layer-name> <batch> <pass: fwd|bwd|dw> <core: 1|2|3> [iters]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd.models.inception_train import InceptionTrainer

name, B, which, core = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
tr = InceptionTrainer(5, B, "cuda", conv_core="v2")
x = torch.zeros(B, 299, 299, 8, dtype=torch.bfloat16, device="cuda")
op = [o for o in tr.graph.ops if o.kind == "conv" and o.conv.name == name][0]
src = x if op.src == "input" else tr.act[op.src]
src.normal_()
tr.z[name].normal_()
st = torch.cuda.current_stream().cuda_stream
fn = {"fwd": lambda: tr._conv_fwd(op, src, core, st), "dw": lambda: tr._conv_dw(op, src, core, st),
      "bwd": lambda: tr._conv_bwd_data(op, src, core, 0, st)}[which]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
fn()
torch.cuda.synchronize()
ev0.record()
for _ in range(iters):
    fn()
ev1.record()
torch.cuda.synchronize()
s = op.conv
ho, wo, _ = tr.graph.shapes[op.dst]
fl = 2.0 * B * ho * wo * s.cout * s.kh * s.kw * s.cin
us = ev0.elapsed_time(ev1) / iters * 1e3
print("%s %s core %d B=%d: %.1f us, %.1f TFLOP/s" % (name, which, core, B, us, fl / us / 1e6))
