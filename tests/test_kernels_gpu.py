"""Numerics of every HIP kernel vs a plain PyTorch fp32 reference of the same op (GPU only).

bf16 MFMA operands: references round the same operands to bf16 and accumulate in fp32, so the
tolerances only have to absorb accumulation-order differences.
"""
import math

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(t):
    return t.to(torch.bfloat16)


def r(t):
    return t.to(torch.bfloat16).float()


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


@pytest.fixture(scope="module")
def ops():
    from distributed_tensorflow_amd.ops import hip_ops
    return hip_ops


def dev():
    return torch.device("cuda")


@pytest.mark.parametrize("M,splits", [(100, 14), (100, 1), (1, 2), (333, 3)])
def test_fc_forward(ops, M, splits):
    torch.manual_seed(0)
    x = bf(torch.randn(M, 3136, device=dev()))
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    slabs = ops.fc_forward(x, w, splits=splits)
    got = slabs.sum(0)
    ref = x.float() @ r(w)
    assert relerr(got, ref) < 2e-3


def test_fc_forward_f32_bias_relu(ops):
    torch.manual_seed(1)
    x = torch.randn(64, 2048, device=dev())
    w = torch.randn(2048, 6, device=dev()) * 0.05
    b = torch.randn(6, device=dev())
    got = ops.fc_forward(x, w, bias=b, relu=True)[0]
    ref = torch.relu(r(x) @ r(w) + b)
    assert relerr(got, ref) < 2e-3


@pytest.mark.parametrize("splits", [1, 4])
def test_fc_backward_data(ops, splits):
    torch.manual_seed(2)
    dy = bf(torch.randn(100, 1024, device=dev()))
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    got = ops.fc_backward_data(dy, w, splits=splits)
    ref = dy.float() @ r(w).t()
    assert relerr(got, ref) < 2e-3


@pytest.mark.parametrize("M,N,Kb", [(3136, 1024, 100), (1024, 10, 100), (64, 6, 7), (2048, 6, 100)])
def test_fc_backward_weight(ops, M, N, Kb):
    torch.manual_seed(3)
    x = bf(torch.randn(Kb, M, device=dev()))
    dy = bf(torch.randn(Kb, N, device=dev()))
    got = ops.fc_backward_weight(x, dy)
    ref = x.float().t() @ dy.float()
    assert relerr(got, ref) < 2e-3


def _conv1_ref(x, w, b):
    c = F.conv2d(x.view(-1, 1, 28, 28), w.permute(3, 2, 0, 1), b, padding=2)
    return c


def test_conv1_forward(ops):
    torch.manual_seed(4)
    imgs = torch.rand(300, 784, device=dev())
    idx = torch.randperm(300, device=dev())[:100].to(torch.int32)
    w = torch.randn(5, 5, 1, 32, device=dev()) * 0.1
    b = torch.randn(32, device=dev()) * 0.1
    out, amax = ops.conv1_forward(imgs, w.view(25, 32).contiguous(), b, index=idx)
    c = _conv1_ref(imgs[idx.long()], w, b)
    ref = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2  # bf16 output rounding
    # argmax points at a maximal element of each window
    cw = c.permute(0, 2, 3, 1).reshape(100, 14, 2, 14, 2, 32).permute(0, 1, 3, 5, 2, 4).reshape(100, 14, 14, 32, 4)
    picked = torch.gather(cw, 4, amax.long().unsqueeze(-1)).squeeze(-1)
    assert torch.allclose(picked, cw.max(-1).values, atol=1e-5)


def test_conv1_forward_no_index(ops):
    torch.manual_seed(5)
    imgs = torch.rand(7, 784, device=dev())
    w = torch.randn(5, 5, 1, 32, device=dev()) * 0.1
    b = torch.zeros(32, device=dev())
    out, _ = ops.conv1_forward(imgs, w.view(25, 32).contiguous(), b)
    ref = F.max_pool2d(torch.relu(_conv1_ref(imgs, w, b)), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2


def test_conv1_forward_pool_ties_and_dead_windows(ops):
    """Flat images make every interior 2x2 window a four-way tie; the argmax must still name ONE position of the
    window, and un-pooling through it must conserve the gradient.  A negative bias on half of the channels makes those
    windows all-zero after ReLU: they pool to zero and take no gradient."""
    B = 4
    imgs = torch.full((B, 784), 0.5, device=dev())
    imgs[1].zero_()
    w = torch.full((5, 5, 1, 32), 0.04, device=dev())
    b = torch.cat([torch.full((16,), 0.1), torch.full((16,), -5.0)]).to(dev())
    out, amax = ops.conv1_forward(imgs, w.view(25, 32).contiguous(), b)
    ref = F.max_pool2d(torch.relu(_conv1_ref(imgs, w, b)), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2
    assert int(amax.max()) <= 3
    assert float(out[..., 16:].float().abs().max()) == 0.0
    dpool = torch.rand(B, 14, 14, 32, device=dev()) + 0.5
    dy, db = ops.unpool(dpool, out, amax)
    live = (out.float() > 0)
    assert relerr(dy.float().reshape(B, 14, 2, 14, 2, 32).sum((2, 4)), dpool * live) < 1e-2
    # exactly one position per live window carries the gradient
    nz = (dy.float().reshape(B, 14, 2, 14, 2, 32) != 0).sum((2, 4))
    assert torch.equal(nz, live.long())
    assert relerr(db, (dpool * live).sum((0, 1, 2))) < 1e-4


def test_conv_forward_at_evaluation_batch(ops):
    """The evaluation path pushes the whole 10 000-image test set through the forward kernels in one call."""
    torch.manual_seed(21)
    B = 10000
    imgs = torch.rand(B, 784, device=dev())
    w1 = torch.randn(5, 5, 1, 32, device=dev()) * 0.1
    b1 = torch.randn(32, device=dev()) * 0.1
    p1, _ = ops.conv1_forward(imgs, w1.view(25, 32).contiguous(), b1)
    ref1 = F.max_pool2d(torch.relu(_conv1_ref(imgs, w1, b1)), 2).permute(0, 2, 3, 1)
    assert relerr(p1, ref1) < 1e-2
    w2 = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    b2 = torch.randn(64, device=dev()) * 0.1
    p2, _ = ops.conv_bias_relu_pool(p1, w2.view(800, 64).contiguous(), b2, 5, 5)
    c = F.conv2d(p1.float().permute(0, 3, 1, 2), r(w2).permute(3, 2, 0, 1), b2, padding=2)
    ref2 = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    assert relerr(p2, ref2) < 1e-2


@pytest.mark.parametrize("B", [100, 3])
def test_conv2_forward_pool(ops, B):
    torch.manual_seed(6)
    x = bf(torch.rand(B, 14, 14, 32, device=dev()))
    w = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    out, amax = ops.conv_bias_relu_pool(x, w.view(800, 64).contiguous(), b, 5, 5)
    c = F.conv2d(x.float().permute(0, 3, 1, 2), r(w).permute(3, 2, 0, 1), b, padding=2)
    ref = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2
    cw = c.permute(0, 2, 3, 1).reshape(B, 7, 2, 7, 2, 64).permute(0, 1, 3, 5, 2, 4).reshape(B, 7, 7, 64, 4)
    picked = torch.gather(cw, 4, amax.long().unsqueeze(-1)).squeeze(-1)
    assert float((picked - cw.max(-1).values).abs().max()) < 2e-2


def test_conv_pool_generic_shape(ops):
    """A shape that does not match the LDS-resident MNIST kernel goes through the generic implicit GEMM."""
    torch.manual_seed(16)
    B = 5
    x = bf(torch.rand(B, 10, 12, 16, device=dev()))
    w = torch.randn(3, 3, 16, 32, device=dev()) * 0.1
    b = torch.randn(32, device=dev()) * 0.1
    out, amax = ops.conv_bias_relu_pool(x, w.view(144, 32).contiguous(), b, 3, 3)
    c = F.conv2d(x.float().permute(0, 3, 1, 2), r(w).permute(3, 2, 0, 1), b, padding=1)
    ref = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-2


def test_unpool(ops):
    torch.manual_seed(7)
    B = 10
    dpool = torch.randn(B, 7, 7, 64, device=dev())
    pooled = bf(torch.relu(torch.randn(B, 7, 7, 64, device=dev())))
    amax = torch.randint(0, 4, (B, 7, 7, 64), device=dev(), dtype=torch.uint8)
    dy, db = ops.unpool(dpool, pooled, amax)
    g = dpool * (pooled.float() > 0)
    ref = torch.zeros(B, 7, 2, 7, 2, 64, device=dev())
    for s in range(4):
        ref[:, :, s >> 1, :, s & 1, :] = g * (amax == s)
    ref = ref.reshape(B, 14, 14, 64)
    assert relerr(dy, ref) < 1e-2
    assert relerr(db, g.sum((0, 1, 2))) < 1e-4


def test_conv_backward_data_mask(ops):
    torch.manual_seed(8)
    B = 20
    dy = bf(torch.randn(B, 14, 14, 64, device=dev()))
    w = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    mask = bf(torch.relu(torch.randn(B, 14, 14, 32, device=dev())))
    got = ops.conv_backward_data_mask(dy, w.view(800, 64).contiguous(), mask, 5, 5)
    x = torch.zeros(B, 32, 14, 14, device=dev(), requires_grad=True)
    y = F.conv2d(x, r(w).permute(3, 2, 0, 1), padding=2)
    (gx,) = torch.autograd.grad(y, x, dy.float().permute(0, 3, 1, 2))
    ref = gx.permute(0, 2, 3, 1) * (mask.float() > 0)
    assert relerr(got, ref) < 2e-3


@pytest.mark.parametrize("B,splits", [(100, 32), (5, 1)])
def test_conv_backward_weight(ops, B, splits):
    torch.manual_seed(9)
    x = bf(torch.rand(B, 14, 14, 32, device=dev()))
    dy = bf(torch.randn(B, 14, 14, 64, device=dev()))
    got = ops.conv_backward_weight(x, dy, 5, 5, splits=splits)
    wt = torch.zeros(64, 32, 5, 5, device=dev(), requires_grad=True)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), wt, padding=2)
    (gw,) = torch.autograd.grad(y, wt, dy.float().permute(0, 3, 1, 2))
    ref = gw.permute(2, 3, 1, 0)
    assert relerr(got, ref) < 2e-3


@pytest.mark.parametrize("B,ipw", [(100, 4), (7, 3)])
def test_conv2_backward_lds_matches_unfused_chain(ops, B, ipw):
    torch.manual_seed(20 + B)
    pool1 = bf(torch.relu(torch.randn(B, 14, 14, 32, device=dev())))
    w = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    pool2, amax2 = ops.conv_bias_relu_pool(pool1, w.view(800, 64).contiguous(), b, 5, 5)
    dpool2 = torch.randn(B, 7, 7, 64, device=dev())
    dpool1, dW, db = ops.conv2_backward_lds(dpool2, pool2, amax2, w.contiguous(), pool1, images_per_block=ipw)
    # reference: explicit un-pool, then PyTorch conv gradients on the same bf16-rounded operands
    g = dpool2 * (pool2.float() > 0)
    dy = torch.zeros(B, 7, 2, 7, 2, 64, device=dev())
    for s_ in range(4):
        dy[:, :, s_ >> 1, :, s_ & 1, :] = g * (amax2 == s_)
    dy = r(dy.reshape(B, 14, 14, 64))
    xin = pool1.float().permute(0, 3, 1, 2).requires_grad_(True)
    wt = r(w).permute(3, 2, 0, 1).requires_grad_(True)
    y = F.conv2d(xin, wt, padding=2)
    gx, gw = torch.autograd.grad(y, (xin, wt), dy.permute(0, 3, 1, 2))
    ref_dx = gx.permute(0, 2, 3, 1) * (pool1.float() > 0)
    assert relerr(dpool1, ref_dx) < 2e-3
    assert relerr(dW, gw.permute(2, 3, 1, 0)) < 2e-3
    assert relerr(db, g.sum((0, 1, 2))) < 1e-4


def test_conv1_backward_weight(ops):
    torch.manual_seed(10)
    B = 50
    imgs = torch.rand(B, 784, device=dev())
    w = torch.randn(5, 5, 1, 32, device=dev()) * 0.1
    b = torch.randn(32, device=dev()) * 0.1
    out, amax = ops.conv1_forward(imgs, w.view(25, 32).contiguous(), b)
    dpool = torch.randn(B, 14, 14, 32, device=dev()) * (out.float() > 0)
    dW, db = ops.conv1_backward_weight(imgs, dpool.contiguous(), amax)
    wt = w.clone().requires_grad_(True)
    bt = b.clone().requires_grad_(True)
    ref_out = F.max_pool2d(torch.relu(_conv1_ref(imgs, wt, bt)), 2).permute(0, 2, 3, 1)
    gw, gb = torch.autograd.grad(ref_out, (wt, bt), dpool)
    assert relerr(dW.view(5, 5, 1, 32), gw) < 1e-3
    assert relerr(db, gb) < 1e-3


def test_adam_tf(ops):
    from distributed_tensorflow_amd.optim.adam_tf import adam_tf_flat_
    torch.manual_seed(11)
    n = 4096 * 3 + 64
    P = torch.randn(n, device=dev())
    M = torch.zeros(n, device=dev())
    V = torch.zeros(n, device=dev())
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    ticket = torch.zeros(1, dtype=torch.int32, device=dev())
    Pr, Mr, Vr = P.clone(), M.clone(), V.clone()
    for t in range(1, 4):
        G = torch.randn(n, device=dev())
        Gk = G.clone()
        ops.adam_tf_(P, Gk, M, V, step, ticket, lr=1e-3, grad_scale=0.5, zero_from=1024)
        adam_tf_flat_(Pr, G, Mr, Vr, t, 1e-3, grad_scale=0.5)
        torch.cuda.synchronize()
        assert int(step.item()) == t and int(ticket.item()) == 0
        assert float(Gk[1024:].abs().max()) == 0.0 and torch.equal(Gk[:1024], G[:1024])
        assert torch.allclose(P, Pr, rtol=1e-5, atol=1e-6)
        assert torch.allclose(V, Vr, rtol=1e-4, atol=1e-9)


def test_sgd(ops):
    torch.manual_seed(12)
    n = 2048 * 5 + 12
    P = torch.randn(n, device=dev())
    G = torch.randn(n, device=dev())
    ref = P - 0.01 * G
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    ticket = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.sgd_(P, G, step, ticket, lr=0.01)
    torch.cuda.synchronize()
    assert torch.allclose(P, ref, rtol=1e-6, atol=1e-7) and float(G.abs().max()) == 0.0 and int(step.item()) == 1


def test_tensor_stats_kernel_matches_numpy():
    from distributed_tensorflow_amd.utils import summary
    torch.manual_seed(3)
    parts = [torch.randn(70001) * 0.03, torch.zeros(17), torch.tensor([1e-12, -1e-12, 1.1e-12, 5e19, -3e25, 1.0, -1.0]),
             torch.randn(3000).exp() * 50, torch.tensor([float("nan")])]
    x = torch.cat(parts).to(dev())
    mn, mx, num, sm, ss, hist = summary.device_stats(x)
    a = x.cpu().numpy().astype(np.float64)
    a = a[~np.isnan(a)]
    ref = np.bincount(np.minimum(np.searchsorted(summary.BUCKET_LIMITS, a, side="right"), len(summary.BUCKET_LIMITS) - 1),
                      minlength=len(summary.BUCKET_LIMITS))
    assert num == a.size and mn == a.min() and mx == a.max()
    assert abs(sm - a.sum()) <= 1e-9 * np.abs(a).sum() and abs(ss - (a * a).sum()) <= 1e-9 * (a * a).sum()
    assert np.array_equal(hist, ref)
    s = summary.variable_summaries("w", x[:70001])
    d = dict(s.scalars)
    b = a[:70001]
    assert abs(d["w/summaries/mean"] - b.mean()) < 1e-7 and abs(d["w/summaries/stddev/stddev"] - b.std()) < 1e-6
    assert d["w/summaries/max"] == b.max() and d["w/summaries/min"] == b.min()
    assert sum(s.histograms[0][1]["bucket"]) == b.size
