"""The full native Inception fine-tune step vs the autograd oracle of the whole network, layer by layer (GPU only).

Kept in a file of its own that sorts after the kernel-level suites: it is the longest single GPU test and the only one whose
margins depend on 47 layers of accumulated bf16 rounding, so the per-kernel tests report first."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _r(t):
    return t.to(BF).float()


def _relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bf16: the native backward stores dz (the gradient of the conv output) in bf16
    and both the filter- and the data-gradient GEMM read that, so the oracle rounds at the same place."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(BF).float()


def _l2err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def test_full_network_step_matches_autograd_layerwise():
    """A random-init 47-layer BatchNorm net amplifies bf16 rounding noise ~1.5x per block (measured: exact in the stem,
    4% at mixed_4, 30% at mixed_10 against the bf16-simulating oracle), so the end-to-end comparison is only a loose
    sanity check; the rigorous check is teacher-forced: every conv's forward and all of its gradients are recomputed
    with autograd from the native executor's own input activation, upstream gradient and ReLU mask, with the gradient of
    the conv output rounded to bf16 where the native backward stores it.  (The same inputs and cores do not give the same
    bits twice: the fp32 atomics of the BN statistics reorder, 76 of 2.8 M first-layer outputs round the other way, and
    bf16 rounding turns a perturbation eps into sqrt(eps * ulp): 3e-5 after conv, 2e-3 after conv_4, 0.46 at mixed_10,
    scripts/dbg/whole_network_determinism.py.  Without the forced mask and rounding the margins were random variables
    of that noise and one run in ~40 crossed a 2e-2 bound on a beta gradient.)"""
    from distributed_tensorflow_amd.models.inception_train import InceptionTrainer, torch_train_step
    from distributed_tensorflow_amd.models.inception_v3 import BN_EPS
    torch.manual_seed(0)
    B, C = 4, 5
    tr = InceptionTrainer(C, B, "cuda", lr=0.01, seed=3)
    x = torch.zeros(B, 299, 299, 8, device="cuda")
    x[..., :3] = torch.randn(B, 299, 299, 3, device="cuda")
    x = x.to(BF)
    labels = torch.tensor([0, 3, 1, 4], device="cuda", dtype=torch.int32)
    params = {k: tr.layout.view(tr.P, k).clone() for k in tr.layout.entries}
    tr.fwd_bwd(x, labels)
    torch.cuda.synchronize()
    loss_ref, grads, feat_ref = torch_train_step(tr.graph, params, x[..., :3].float().permute(0, 3, 1, 2).contiguous(), labels,
                                                 bf16_sim=True)
    loss, acc = tr.metrics()
    cos_feat, dloss = _cos(tr.feat, feat_ref), abs(loss - float(loss_ref))
    print("end to end: cos(features) %.5f, |loss - oracle| %.5f" % (cos_feat, dloss))
    assert cos_feat > 0.98, cos_feat
    assert dloss < 5e-2, dloss
    # head gradients from the native features
    W = params["final_training_ops/weights/final_weights"].clone().requires_grad_(True)
    b = params["final_training_ops/biases/final_biases"].clone().requires_grad_(True)
    f = tr.feat.clone().requires_grad_(True)
    l = F.cross_entropy(f @ W + b, labels.long())
    gW, gb_, gf = torch.autograd.grad(l, (W, b, f))
    assert _relerr(tr.grad("final_training_ops/weights/final_weights"), gW) < 1e-3
    assert _relerr(tr.grad("final_training_ops/biases/final_biases"), gb_) < 1e-3
    assert _relerr(tr.dfeat, gf) < 1e-3
    consumers = {}
    for op in tr.graph.ops:
        consumers[op.src] = consumers.get(op.src, 0) + 1
    worst = {"y": 0.0, "dW": 0.0, "dgamma": 0.0, "dbeta": 0.0, "dx": 0.0}
    worst_at = {}
    # 70 repeats with fresh autotunes (scripts/dbg/whole_network_repeat.py): y <= 1.73e-3 and dx <= 1.67e-3 (the bf16 rounding of
    # the stored tensor itself), dW <= 4.6e-4, dgamma <= 3.9e-4, dbeta <= 1.5e-6
    tol = {"y": 5e-3, "dx": 5e-3, "dW": 3e-3, "dgamma": 3e-3, "dbeta": 1e-3}
    n_dx = 0
    for op in tr.graph.ops:
        if op.kind != "conv":
            continue
        s = op.conv
        src = x if op.src == "input" else tr.act[op.src]
        xin = src.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        w = _r(params[s.name + "/conv2d_params"]).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
        gm = params[s.name + "/batchnorm/gamma"].clone().requires_grad_(True)
        bt = params[s.name + "/batchnorm/beta"].clone().requires_grad_(True)
        z = _RoundGrad.apply(F.conv2d(xin, w, stride=s.stride, padding=s.pad))
        z = z + (_r(z) - z).detach()
        mean, var = z.mean((0, 2, 3), keepdim=True), z.var((0, 2, 3), unbiased=False, keepdim=True)
        pre = (z - mean) * torch.rsqrt(var + BN_EPS) * gm.view(1, -1, 1, 1) + bt.view(1, -1, 1, 1)
        y = F.relu(pre)
        y_nat = tr.act[op.dst][..., op.dst_off:op.dst_off + s.cout].float().permute(0, 3, 1, 2)
        dy = tr.dact[op.dst][..., op.dst_off:op.dst_off + s.cout].float().permute(0, 3, 1, 2)
        # the ReLU mask is teacher-forced too: the oracle's batch mean is that of the bf16-rounded conv output, the native
        # one that of the fp32 accumulators, so ~5 activations a layer with |y| < 3e-5 land on the other side of zero; beta's
        # gradient is a cancelling sum of dy over the mask, and one flipped element with a large dy moved its relative
        # error from the usual 0.5 % to 2.4 % in about one run in forty (which conv cores the autotuner picked decides which
        # elements flip).  The forward check on y below still sees a wrong mask
        gx, gw, gg, gb = torch.autograd.grad(pre * (y_nat > 0).float(), (xin, w, gm, bt), dy)
        errs = {"y": _l2err(y_nat, y.detach()), "dW": _l2err(tr.grad(s.name + "/conv2d_params"), gw.permute(2, 3, 1, 0)),
                "dgamma": _l2err(tr.grad(s.name + "/batchnorm/gamma"), gg), "dbeta": _l2err(tr.grad(s.name + "/batchnorm/beta"), gb)}
        if op.src != "input" and consumers[op.src] == 1:
            errs["dx"] = _l2err(tr.dact[op.src], gx.permute(0, 2, 3, 1))
            n_dx += 1
        for k, v in errs.items():
            if v > worst[k]:
                worst[k], worst_at[k] = v, s.name
            assert v < tol[k], (s.name, k, v)
    print("worst layerwise relative L2 errors:", worst)
    print("at:", {k: (n, tr.core[n]) for k, n in worst_at.items()})
    assert n_dx > 40
    # branch points: the gradient of a block input is the sum over its four consumers
    for blk_src, blk in (("mixed_8", "mixed_9"), ("pool_1:out", "mixed")):
        xin = tr.act[blk_src].float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        total = torch.zeros_like(xin)
        for op in tr.graph.ops:
            if op.src != blk_src:
                continue
            if op.kind == "conv":
                s = op.conv
                w = _r(params[s.name + "/conv2d_params"]).permute(3, 2, 0, 1).contiguous()
                z = _RoundGrad.apply(F.conv2d(xin, w, stride=s.stride, padding=s.pad))
                z = z + (_r(z) - z).detach()
                mean, var = z.mean((0, 2, 3), keepdim=True), z.var((0, 2, 3), unbiased=False, keepdim=True)
                y = ((z - mean) * torch.rsqrt(var + BN_EPS) * params[s.name + "/batchnorm/gamma"].view(1, -1, 1, 1)
                     + params[s.name + "/batchnorm/beta"].view(1, -1, 1, 1))
                y = y * (tr.act[op.dst][..., op.dst_off:op.dst_off + s.cout].float().permute(0, 3, 1, 2) > 0).float()
                dy = tr.dact[op.dst][..., op.dst_off:op.dst_off + s.cout].float().permute(0, 3, 1, 2)
            elif op.kind == "avgpool":
                y = F.avg_pool2d(xin, 3, 1, 1, count_include_pad=False)
                dy = tr.dact[op.dst].float().permute(0, 3, 1, 2)
            else:
                continue
            total = total + torch.autograd.grad(y, xin, dy)[0]
        e = _l2err(tr.dact[blk_src], total.permute(0, 2, 3, 1))
        print("branch point %s: relative L2 error of the summed gradient %.2e" % (blk_src, e))
        assert e < 8e-3, blk_src  # 2.65e-3 in every one of 40 runs: bf16 roundings of the accumulated gradient
    # the optimizer step consumes and clears the gradient arena
    p0 = tr.P.clone()
    g0 = tr.G.clone()
    tr.apply()
    torch.cuda.synchronize()
    assert torch.allclose(tr.P, p0 - 0.01 * g0, atol=1e-6) and float(tr.G.abs().max()) == 0.0
    assert int(tr.step_t.item()) == 1
