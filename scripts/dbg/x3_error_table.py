"""Error of every fp32 kernel against a float64 oracle, exact-fp32 MFMA vs bf16x6 (DTF_F32_EMULATION), same inputs.
Run on a GPU box: python scripts/dbg/x3_error_table.py"""
import torch
import torch.nn.functional as F
from distributed_tensorflow_amd import native
from distributed_tensorflow_amd.ops import hip_ops as ops

dev = torch.device("cuda")


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def rms(a, b):
    return float(((a.double() - b.double()) ** 2).mean().sqrt() / ((b.double() ** 2).mean().sqrt() + 1e-30))


def run(B=100):
    out = {}
    torch.manual_seed(6)
    x = torch.relu(torch.randn(B, 14, 14, 32, device=dev))
    w = torch.randn(5, 5, 32, 64, device=dev) * 0.05
    b = torch.randn(64, device=dev) * 0.1
    pool2, amax2 = ops.conv2_forward_f32(x, w, b)
    c = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), padding=2)
    ref = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    out["conv2 fwd"] = (relerr(pool2, ref), rms(pool2, ref))
    dpool2 = torch.randn(B, 7, 7, 64, device=dev)
    dpool1, dW, db = ops.conv2_backward_f32(dpool2, pool2, amax2, w, x)
    xp = x.double().clone().requires_grad_(True)
    w64 = w.double().clone().requires_grad_(True)
    b64 = b.double().clone().requires_grad_(True)
    c = F.conv2d(torch.relu(xp).permute(0, 3, 1, 2), w64.permute(3, 2, 0, 1), b64, padding=2)
    p2 = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    (p2 * dpool2.double()).sum().backward()
    out["conv2 dX"] = (relerr(dpool1, xp.grad), rms(dpool1, xp.grad))
    out["conv2 dW"] = (relerr(dW, w64.grad), rms(dW, w64.grad))
    out["conv2 db"] = (relerr(db, b64.grad), rms(db, b64.grad))
    X = torch.relu(torch.randn(B, 3136, device=dev))
    W = torch.randn(3136, 1024, device=dev) * 0.05
    y = ops.fc_forward_f32(X, W, splits=14) if hasattr(ops, "fc_forward_f32") else None
    if y is not None:
        y = y.sum(0) if y.dim() == 3 else y
        out["fc1 fwd"] = (relerr(y, X.double() @ W.double()), rms(y, X.double() @ W.double()))
    return out


if __name__ == "__main__":
    lib = native.hip()
    res = {}
    for emu in (0, 1):
        lib.dtf_f32_emulation_set(emu)
        res[emu] = run()
    print("%-12s %12s %12s %12s %12s" % ("kernel", "exact max", "exact rms", "bf16x6 max", "bf16x6 rms"))
    for k in res[0]:
        print("%-12s %12.3e %12.3e %12.3e %12.3e" % (k, res[0][k][0], res[0][k][1], res[1][k][0], res[1][k][1]))
