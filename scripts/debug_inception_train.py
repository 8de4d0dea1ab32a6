"""Per-buffer forward error of the native training forward vs the torch oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.models.inception_train import InceptionTrainer, torch_forward_train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
tr = InceptionTrainer(5, B, "cuda", seed=3)
x = torch.zeros(B, 299, 299, 8, device="cuda")
x[..., :3] = torch.randn(B, 299, 299, 3, device="cuda")
x = x.to(torch.bfloat16)
params = {k: tr.layout.view(tr.P, k).clone() for k in tr.layout.entries}
tr.forward(x)
torch.cuda.synchronize()
for sim in (True, False):
    with torch.no_grad():
        feat, bufs = torch_forward_train(tr.graph, params, x[..., :3].float().permute(0, 3, 1, 2).contiguous(), bf16_sim=sim,
                                         return_buffers=True)
    print("bf16_sim", sim)
    for name, ref in bufs.items():
        got = tr.act[name].float().permute(0, 3, 1, 2)
        err = float((got - ref).abs().mean() / (ref.abs().mean() + 1e-12))
        print("  %-40s mean-rel-err %.4f" % (name, err))
    print("  feat err %.4f" % float((tr.feat - feat).abs().mean() / feat.abs().mean()))
