"""Reference-precision (fp32) path: every exact-fp32 MFMA kernel and the whole native step against an UN-ROUNDED
oracle (float64 PyTorch of the same op / model), GPU only.

The reference computes in float32 (``/root/reference/demo1/train.py:28-41,103,120,132``); ``v_mfma_f32_16x16x4_f32`` is a
k-ordered fmaf chain, so the only difference from the float64 oracle is fp32 accumulation rounding: the tolerances below are
1e-5 .. 1e-4 relative, two to three orders tighter than the bf16 tests.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from distributed_tensorflow_amd.ops import hip_ops
    return hip_ops


def dev():
    return torch.device("cuda")


@pytest.mark.parametrize("M,splits", [(100, 14), (100, 1), (1, 2), (333, 3), (5000, 1)])
def test_fc_forward_f32(ops, M, splits):
    torch.manual_seed(0)
    x = torch.randn(M, 3136, device=dev())
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    got = ops.fc_forward_f32(x, w, splits=splits).double().sum(0)
    ref = x.double() @ w.double()
    assert relerr(got, ref) < 1e-5


def test_fc_forward_f32_bias_relu_asymmetric(ops):
    """Identity-like A against an asymmetric B catches a transposed fragment map."""
    K, N = 64, 128
    x = torch.eye(K, device=dev())[:48].contiguous()
    w = (torch.arange(K * N, device=dev(), dtype=torch.float32).reshape(K, N) % 251) - 100.0
    b = torch.arange(N, device=dev(), dtype=torch.float32) * 0.5
    got = ops.fc_forward_f32(x, w, bias=b, relu=True)[0]
    ref = torch.relu(w[:48] + b)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("splits", [1, 4])
def test_fc_backward_data_f32(ops, splits):
    torch.manual_seed(2)
    dy = torch.randn(100, 1024, device=dev())
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    got = ops.fc_backward_data_f32(dy, w, splits=splits)
    ref = dy.double() @ w.double().t()
    assert relerr(got, ref) < 1e-5


@pytest.mark.parametrize("M,N,Kb,ldy", [(3136, 1024, 100, 1024), (1024, 10, 100, 16), (64, 6, 7, 8), (2048, 64, 37, 64)])
def test_fc_backward_weight_f32(ops, M, N, Kb, ldy):
    torch.manual_seed(3)
    x = torch.randn(Kb, M, device=dev())
    dy = torch.zeros(Kb, ldy, device=dev())
    dy[:, :N] = torch.randn(Kb, N, device=dev())
    got = ops.fc_backward_weight_f32(x, dy, n_valid=N)
    ref = x.double().t() @ dy[:, :N].double()
    assert got.shape == (M, N) and relerr(got, ref) < 2e-6


@pytest.mark.parametrize("granks", [0, 2, 3, 8])
def test_fc_backward_trio_f32(granks):
    """The fused fc backward launch (fc1 data gradient | fc1 dW | fc2 dW); with ``granks`` > 0 the fc1 filter gradient is
    summed over the ranks' operand blocks (the "gather" data-parallel schedule), blocks at unrelated addresses."""
    import ctypes as C
    from distributed_tensorflow_amd import native
    lib = native.hip()
    P, I = C.c_void_p, C.c_int
    lib.dtf_fc_bwd_trio_f32.argtypes = [P, P, P, I, I, P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, P]
    lib.dtf_fc_bwd_trio_f32.restype = I
    torch.manual_seed(5 + granks)
    B, n = 100, max(1, granks)
    xs = [torch.randn(B, 3136, device=dev()) for _ in range(n)]
    pad = torch.empty(4099, device=dev())  # (keeps the blocks from being evenly spaced)
    dhs = [torch.randn(B, 1024, device=dev()) for _ in range(n)]
    hdrop = torch.randn(B, 1024, device=dev())
    dz = torch.zeros(B, 16, device=dev())
    dz[:, :10] = torch.randn(B, 10, device=dev())
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    dW0 = torch.full((3136, 1024), float("nan"), device=dev())
    dW1 = torch.full((1024, 10), float("nan"), device=dev())
    dX = torch.zeros(B, 3136, device=dev())
    ga = (P * 8)(*[x.data_ptr() for x in xs])
    gb = (P * 8)(*[d.data_ptr() for d in dhs])
    torch.cuda.synchronize()
    rc = lib.dtf_fc_bwd_trio_f32(xs[0].data_ptr(), dhs[0].data_ptr(), dW0.data_ptr(), 3136, 1024, hdrop.data_ptr(), dz.data_ptr(),
                                 16, dW1.data_ptr(), 1024, 10, B, dhs[0].data_ptr(), w.data_ptr(), dX.data_ptr(), B, 3136, 1024,
                                 8, ga if granks else None, gb if granks else None, granks, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref0 = sum(x.double().t() @ d.double() for x, d in zip(xs, dhs))
    assert relerr(dW0, ref0) < 2e-6
    assert relerr(dW1, hdrop.double().t() @ dz[:, :10].double()) < 2e-6
    assert relerr(dX, dhs[0].double() @ w.double().t()) < 1e-5
    # the same launch with the bias gradients as column sums of dYq / dY1 (what the engine uses: the loss head then adds none)
    lib.dtf_fc_bwd_trio_f32_bias.argtypes = [P, P, P, I, I, P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, P, P, P, I, P, P]
    lib.dtf_fc_bwd_trio_f32_bias.restype = I
    db0 = torch.full((1024,), float("nan"), device=dev())
    db1 = torch.full((10,), float("nan"), device=dev())
    dz[:, 12:14] = torch.rand(B, 2, device=dev())   # (per-row loss / hit count, as the loss head leaves them)
    ring = torch.zeros(8, 2, device=dev())
    ring[3] = torch.tensor([0.5, 2.0])
    step = torch.tensor([11], dtype=torch.int32, device=dev())   # slot 11 % 8 = 3
    dW0.fill_(float("nan")); dW1.fill_(float("nan")); dX.zero_()
    torch.cuda.synchronize()
    rc = lib.dtf_fc_bwd_trio_f32_bias(xs[0].data_ptr(), dhs[0].data_ptr(), dW0.data_ptr(), 3136, 1024, hdrop.data_ptr(),
                                      dz.data_ptr(), 16, dW1.data_ptr(), 1024, 10, B, dhs[0].data_ptr(), w.data_ptr(),
                                      dX.data_ptr(), B, 3136, 1024, 8, ga if granks else None, gb if granks else None, granks,
                                      db0.data_ptr(), db1.data_ptr(), ring.data_ptr(), 8, step.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert relerr(db0, dhs[0].double().sum(0)) < 1e-6 and relerr(db1, dz[:, :10].double().sum(0)) < 1e-6
    want = torch.zeros(8, 2, dtype=torch.float64, device=dev())
    want[3] = torch.tensor([0.5, 2.0], dtype=torch.float64, device=dev()) + dz[:, 12:14].double().sum(0)
    assert relerr(ring, want) < 1e-6 and float(ring[[0, 1, 2, 4, 5, 6, 7]].abs().max()) == 0.0
    assert relerr(dW0, ref0) < 2e-6 and relerr(dX, dhs[0].double() @ w.double().t()) < 1e-5
    assert relerr(dW1, hdrop.double().t() @ dz[:, :10].double()) < 2e-6
    # a batch whose K slabs would straddle two ranks' blocks is refused, not mis-summed
    if granks:
        assert lib.dtf_fc_bwd_trio_f32(xs[0].data_ptr(), dhs[0].data_ptr(), dW0.data_ptr(), 3136, 1024, hdrop.data_ptr(),
                                       dz.data_ptr(), 16, dW1.data_ptr(), 1024, 10, 50, dhs[0].data_ptr(), w.data_ptr(),
                                       dX.data_ptr(), 50, 3136, 1024, 8, ga, gb, granks, None) == -22
    del pad


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 1), (3, 1), (4, 3), (8, 0), (8, 5)])
def test_fc_backward_trio_f32_shard(world, rank):
    """The fused fc backward launch of the "shard" data-parallel schedule: problem 0 is this rank's ROW SLICE of the global
    fc1 filter gradient, summed over all ranks' operand blocks (one workgroup set per source rank, atomics into the
    pre-zeroed slice); rows outside the slice are not touched.  Reference: float64."""
    import ctypes as C
    from distributed_tensorflow_amd import native
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    lib = native.hip()
    P, I = C.c_void_p, C.c_int
    lib.dtf_fc_bwd_trio_f32_shard.argtypes = [P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, I, I, P, I, I, P, P, P]
    lib.dtf_fc_bwd_trio_f32_shard.restype = I
    torch.manual_seed(50 + 8 * world + rank)
    B = 100
    lo, hi = NativeMnistEngine.shard_rows(rank, world)
    assert lo % 4 == 0 and hi % 4 == 0 and (rank < world - 1 or hi == 3136) and (rank > 0 or lo == 0)
    xs = [torch.randn(B, 3136, device=dev()) for _ in range(world)]
    dhs = [torch.randn(B, 1024, device=dev()) for _ in range(world)]
    hdrop = torch.randn(B, 1024, device=dev())
    dz = torch.zeros(B, 16, device=dev())
    dz[:, :10] = torch.randn(B, 10, device=dev())
    w = torch.randn(3136, 1024, device=dev()) * 0.05
    dW0 = torch.zeros(3136, 1024, device=dev())
    if world <= 2:
        dW0[lo:hi] = float("nan")   # (one K set: plain stores, nothing to pre-zero)
    dW0[:lo] = 7.0   # (other ranks' rows: must stay as they are)
    dW0[hi:] = 7.0
    dW1 = torch.full((1024, 10), float("nan"), device=dev())
    dX = torch.zeros(B, 3136, device=dev())
    ga = (P * 8)(*[x.data_ptr() + 4 * lo for x in xs])   # every block already offset to the slice's first column
    gb = (P * 8)(*[d.data_ptr() for d in dhs])
    torch.cuda.synchronize()
    rc = lib.dtf_fc_bwd_trio_f32_shard(hdrop.data_ptr(), dz.data_ptr(), 16, dW1.data_ptr(), 1024, 10, B, dhs[rank].data_ptr(),
                                       w.data_ptr(), dX.data_ptr(), B, 3136, 1024, 6, ga, gb, world, 2, 3136,
                                       dW0.data_ptr() + 4 * lo * 1024, hi - lo, 1024, None, None, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref0 = sum(x[:, lo:hi].double().t() @ d.double() for x, d in zip(xs, dhs))
    assert relerr(dW0[lo:hi], ref0) < 2e-6
    assert bool((dW0[:lo] == 7.0).all()) and bool((dW0[hi:] == 7.0).all())
    assert relerr(dW1, hdrop.double().t() @ dz[:, :10].double()) < 2e-6
    assert relerr(dX, dhs[rank].double() @ w.double().t()) < 1e-5
    # "wshard": the data gradient's weight operand is double-buffered by the parity of a device word
    w2 = torch.randn(3136, 1024, device=dev()) * 0.05
    for word, wexp in ((6, w), (9, w2)):
        sel = torch.tensor([word], dtype=torch.int32, device=dev())
        dX.zero_()
        dW0[lo:hi] = 0.0
        torch.cuda.synchronize()
        rc = lib.dtf_fc_bwd_trio_f32_shard(hdrop.data_ptr(), dz.data_ptr(), 16, dW1.data_ptr(), 1024, 10, B, dhs[rank].data_ptr(),
                                           w.data_ptr(), dX.data_ptr(), B, 3136, 1024, 6, ga, gb, world, 2, 3136,
                                           dW0.data_ptr() + 4 * lo * 1024, hi - lo, 1024, w2.data_ptr(), sel.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert relerr(dX, dhs[rank].double() @ wexp.double().t()) < 1e-5
        assert relerr(dW0[lo:hi], ref0) < 2e-6


def _conv2_ref(x, w, b):
    c = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), padding=2)
    return c, F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)


@pytest.mark.parametrize("B", [100, 3, 1])
def test_conv2_forward_f32(ops, B):
    torch.manual_seed(6)
    x = torch.rand(B, 14, 14, 32, device=dev())
    w = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    out, amax = ops.conv2_forward_f32(x, w, b)
    c, ref = _conv2_ref(x, w, b)
    assert relerr(out, ref) < 1e-5
    cw = c.permute(0, 2, 3, 1).reshape(B, 7, 2, 7, 2, 64).permute(0, 1, 3, 5, 2, 4).reshape(B, 7, 7, 64, 4)
    picked = torch.gather(cw, 4, amax.long().unsqueeze(-1)).squeeze(-1)
    assert float((picked - cw.max(-1).values).abs().max()) < 1e-5


@pytest.mark.parametrize("B", [100, 6, 1])
def test_conv2_backward_f32(ops, B):
    torch.manual_seed(9)
    pool1 = torch.relu(torch.randn(B, 14, 14, 32, device=dev()))
    w = torch.randn(5, 5, 32, 64, device=dev()) * 0.05
    b = torch.randn(64, device=dev()) * 0.1
    pool2, amax2 = ops.conv2_forward_f32(pool1, w, b)
    dpool2 = torch.randn(B, 7, 7, 64, device=dev())
    dpool1, dW, db = ops.conv2_backward_f32(dpool2, pool2, amax2, w, pool1)
    # float64 autograd through the same ops; the ReLU of the layer below is part of the fused kernel's epilogue
    x_pre = pool1.double().clone().requires_grad_(True)
    w64 = w.double().clone().requires_grad_(True)
    b64 = b.double().clone().requires_grad_(True)
    xr = torch.relu(x_pre)
    c = F.conv2d(xr.permute(0, 3, 1, 2), w64.permute(3, 2, 0, 1), b64, padding=2)
    p2 = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    (p2 * dpool2.double()).sum().backward()
    assert relerr(dpool1, x_pre.grad) < 2e-5
    assert relerr(dW, w64.grad) < 2e-5
    assert relerr(db, b64.grad) < 2e-5


def test_conv1_forward_f32(ops):
    torch.manual_seed(4)
    B = 17
    images = torch.rand(64, 784, device=dev())
    idx = torch.randperm(64, device=dev())[:B].to(torch.int32)
    w = torch.randn(25, 32, device=dev()) * 0.1
    b = torch.randn(32, device=dev()) * 0.1
    out, amax = ops.conv1_forward_f32(images, w, b, index=idx)
    x = images[idx.long()].double().reshape(B, 1, 28, 28)
    c = F.conv2d(x, w.double().reshape(5, 5, 1, 32).permute(3, 2, 0, 1), b.double(), padding=2)
    ref = F.max_pool2d(torch.relu(c), 2).permute(0, 2, 3, 1)
    assert relerr(out, ref) < 1e-5


# ----------------------------------------------------------------------------- the whole step

def _dataset(n=2000, seed=0):
    from distributed_tensorflow_amd.data import mnist
    imgs, labels = mnist.synthetic_mnist(n, seed=seed)
    ds = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=seed)
    return mnist.DeviceDataset(ds, "cuda", seed=seed)


@pytest.mark.parametrize("double_softmax,B", [(True, 100), (False, 100), (True, 1), (True, 37), (False, 128), (True, 200)])
def test_step_gradients_match_unrounded_oracle(double_softmax, B):
    """One native fp32 step against float64 autograd of the plain model on the same batch and dropout gate: <= 1e-4
    relative error on every gradient (VERDICT r1 'Next round' 1)."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    from distributed_tensorflow_amd.models import mnist_cnn
    dd = _dataset()
    eng = NativeMnistEngine(dd, batch_size=B, keep_prob=0.7, double_softmax=double_softmax, seed=3, use_graph=False,
                            eval_chunk=500, dtype="fp32")
    params = {k: v.clone() for k, v in eng.param_views().items()}
    eng.fwd_bwd(3)
    eng.stream.synchronize()
    x, y = dd.gather(dd.batch_indices(0, B))
    h_drop = eng.ws["h_drop"].view(-1, 1024)[:B]
    gate = (h_drop != 0).double()
    leaf = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    logits = mnist_cnn.forward_logits(leaf, x.double(), keep_prob=0.7, fc1_gate=gate, training=True, bf16_sim=False)
    loss, pred = mnist_cnn.softmax_xent(logits, y.long(), double_softmax=double_softmax)
    grads = torch.autograd.grad(loss, list(leaf.values()))
    lm, acc = eng.read_metrics(0, 1)
    assert abs(float(lm[0]) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach())))
    assert abs(float(acc[0]) - float((pred.argmax(1) == y.long()).double().mean())) < 1e-6
    g_native = eng.param_views(eng.G)
    for name, g_ref in zip(leaf.keys(), grads):
        err = relerr(g_native[name], g_ref)
        assert err < 1e-4, (name, err)


def test_eval_and_predict_match_unrounded_oracle():
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    from distributed_tensorflow_amd.models import mnist_cnn
    dd = _dataset(n=1300, seed=2)
    eng = NativeMnistEngine(dd, batch_size=100, seed=7, eval_chunk=512, dtype="fp32")
    params = {k: v.double() for k, v in eng.param_views().items()}
    loss, acc = eng.evaluate(dd.images, dd.labels)
    logits = mnist_cnn.forward_logits(params, dd.images.double())
    ref_loss, p = mnist_cnn.softmax_xent(logits, dd.labels.long())
    assert abs(loss - float(ref_loss)) < 1e-5
    probs, pred = eng.predict(dd.images[:37])
    assert relerr(probs, p[:37]) < 1e-5
    assert torch.equal(pred.long(), p[:37].argmax(1))
    assert abs(acc - float((p.argmax(1) == dd.labels.long()).double().mean())) < 1e-3


def test_fp32_graph_replay_matches_unfused_sequence():
    """The fused single-graph fp32 step (W_fc1's Adam riding in the conv filter-gradient launch) against
    fwd_bwd(3) + apply(): same state up to fp32 atomic ordering; the accumulated gradient segments are re-zeroed."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    dd1, dd2 = _dataset(n=4000, seed=3), _dataset(n=4000, seed=3)
    e1 = NativeMnistEngine(dd1, batch_size=100, seed=5, eval_chunk=500, lr=1e-3, double_softmax=False, dtype="fp32")
    e2 = NativeMnistEngine(dd2, batch_size=100, seed=5, eval_chunk=500, lr=1e-3, double_softmax=False, use_graph=False,
                           dtype="fp32")
    assert e1.early_fc1_adam()
    for _ in range(3):
        e1.train_step()
        e2.fwd_bwd(3)
        e2.apply()
        torch.cuda.synchronize()
        dp = float((e1.P - e2.P).abs().max())
        dm = float((e1.M - e2.M).abs().max() / e2.M.abs().max())
        assert dp < 1e-4 and dm < 1e-4, (dp, dm)
    assert int(e1.step_t.item()) == 3 == int(e2.step_t.item())
    lo = e1.layout["b_fc1"][0]
    assert float(e1.G[lo:].abs().max()) == 0.0
    e1.train_steps(40)
    e1.stream.synchronize()
    assert "full*16" in e1._graphs and int(e1.step_t.item()) == 43
    lm, _ = e1.read_metrics(3, 43)
    assert np.isfinite(lm).all() and lm[-5:].mean() < lm[:5].mean()


def test_philox_host_statement_matches_device_dropout():
    """utils/philox.py reproduces the head kernel's dropout mask bit for bit (what lets the eager engine share it)."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    from distributed_tensorflow_amd.utils.philox import dropout_keep_mask
    dd = _dataset(n=500, seed=5)
    eng = NativeMnistEngine(dd, batch_size=64, keep_prob=0.7, seed=12345678901234, use_graph=False, eval_chunk=500, dtype="fp32")
    for step in range(2):
        eng.fwd_bwd(1)
        eng.stream.synchronize()
        h = eng.ws["h_drop"].view(-1, 1024)[:64]
        mask = torch.from_numpy(dropout_keep_mask(64, step, eng.seed, 0.7)).cuda()
        assert float(mask.mean()) > 0.65 and float(mask.mean()) < 0.75
        assert int(((h != 0) & (mask == 0)).sum()) == 0          # nothing survives where the host mask drops
        assert float(((h != 0) | (mask == 0)).float().mean()) > 0.6  # and most kept units are positive activations here
        eng.G.zero_()
        eng.apply()


def test_fp32_trajectory_matches_eager_fp32_on_real_digits(assets_dir):
    """3 000 steps of the reference schedule (batch 100, Adam 1e-4, keep 0.7, double softmax) on the shipped t10k digits
    (8 000 train / 2 000 held out), native fp32 against eager fp32 PyTorch on the same batches and the same dropout
    masks: the loss curves track each other and the test accuracies end within 0.3 points (VERDICT r1 'Next round' 1)."""
    import os
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.engine.mnist_engine import EagerMnistEngine, NativeMnistEngine
    d = os.path.join(assets_dir, "MNIST_data")
    imgs = mnist.read_idx_images(os.path.join(d, "t10k-images-idx3-ubyte.gz"))
    labels = mnist.read_idx_labels(os.path.join(d, "t10k-labels-idx1-ubyte.gz"))
    x, y = mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels)
    xt = torch.from_numpy(x[8000:]).cuda()
    yt = torch.from_numpy(labels[8000:].astype(np.int32)).cuda()
    steps = 3000
    dd_n = mnist.DeviceDataset(mnist.DataSet(x[:8000], y[:8000], seed=11), "cuda", seed=11)
    dd_e = mnist.DeviceDataset(mnist.DataSet(x[:8000], y[:8000], seed=11), "cuda", seed=11)
    nat = NativeMnistEngine(dd_n, batch_size=100, seed=21, lr=1e-4, keep_prob=0.7, eval_chunk=1000, dtype="fp32")
    eag = EagerMnistEngine(dd_e, batch_size=100, seed=21, lr=1e-4, keep_prob=0.7, device="cuda", philox_seed=nat.seed)
    ln, le = [], []
    for s0 in range(0, steps, 500):
        nat.train_steps(500)
        ln.append(nat.read_metrics(s0, s0 + 500)[0])
        for _ in range(500):
            eag.train_step()
        le.append(eag.read_metrics(s0, s0 + 500)[0])
    ln, le = np.concatenate(ln), np.concatenate(le)
    _l, acc_n = nat.evaluate(xt, yt)
    _l, acc_e = eag.evaluate(xt, yt)
    # Before the run-to-run noise sets in the curves agree to rounding (measured 2e-6 over 30 steps); from there two runs of
    # the SAME binary part in discrete events (atomics reorder, a ReLU / max-pool tie falls the other way: 4e-4 .. 2e-3 by
    # step 100, profiles/r3/x3_error_and_noise.txt), so the 100-step bound is that floor, not a precision statement.
    assert np.abs(ln[:30] - le[:30]).max() < 1e-4, np.abs(ln[:30] - le[:30]).max()
    assert np.abs(ln[:100] - le[:100]).max() < 1e-2, np.abs(ln[:100] - le[:100]).max()
    assert abs(ln[-200:].mean() - le[-200:].mean()) < 0.01
    assert acc_n > 0.93 and acc_e > 0.93, (acc_n, acc_e)
    assert abs(acc_n - acc_e) <= 0.003 + 1e-9, (acc_n, acc_e)
