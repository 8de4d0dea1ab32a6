"""Model / optimizer / checkpoint parity tests on CPU (oracle level)."""
import math
import os

import numpy as np
import torch

from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import EagerMnistEngine, arena_layout
from distributed_tensorflow_amd.models import mnist_cnn
from distributed_tensorflow_amd.optim.adam_tf import AdamTF, SGD, adam_tf_flat_
from distributed_tensorflow_amd.utils import checkpoint as ckpt


def _dd(n=400, seed=0, **kw):
    imgs, labels = mnist.synthetic_mnist(n, seed=seed)
    ds = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=seed)
    return mnist.DeviceDataset(ds, "cpu", seed=seed, **kw)


def test_parameter_inventory():
    layout, n = arena_layout()
    assert sum(v[1] for v in layout.values()) == mnist_cnn.NUM_PARAMS == 3274634
    assert list(layout.keys())[0] == "W_fc1" and n % 64 == 0
    p = mnist_cnn.init_params(seed=0)
    assert p["W_conv1"].shape == (5, 5, 1, 32) and p["W_fc1"].shape == (3136, 1024)
    assert float(p["b_conv2"].min()) == float(p["b_conv2"].max()) == np.float32(0.1)
    assert float(p["W_fc1"].abs().max()) <= 0.2 + 1e-6  # truncated at 2 sigma
    assert abs(float(p["W_fc1"].std()) - 0.088) < 0.01  # std of a 2-sigma truncated normal = 0.88 sigma


def test_forward_shapes_and_nhwc_flatten():
    p = mnist_cnn.init_params(seed=1)
    x = torch.rand(3, 784)
    logits, inter = mnist_cnn.forward_logits(p, x, return_intermediates=True)
    assert logits.shape == (3, 10) and inter["pool1"].shape == (3, 32, 14, 14) and inter["flat"].shape == (3, 3136)
    # flatten order is (h, w, c): element [h*7*64 + w*64 + c]
    assert torch.equal(inter["flat"][0, 5 * 64 + 3], inter["pool2"][0, 3, 0, 5])


def test_double_softmax_loss_and_gradient():
    z = torch.tensor([[2.0, -1.0, 0.5], [0.1, 0.2, 0.3]], requires_grad=True)
    y = torch.tensor([0, 2])
    loss, p = mnist_cnn.softmax_xent(z, y, double_softmax=True)
    pp = torch.softmax(z.detach(), 1)
    q = torch.softmax(pp, 1)
    assert abs(float(loss) - float(-(torch.log(q[[0, 1], [0, 2]])).mean())) < 1e-6
    loss.backward()
    onehot = torch.nn.functional.one_hot(y, 3).float()
    g = (q - onehot) / 2
    dz = pp * (g - (g * pp).sum(1, keepdim=True))
    assert torch.allclose(z.grad, dz, atol=1e-6)
    l1, _ = mnist_cnn.softmax_xent(z.detach(), y, double_softmax=False)
    assert abs(float(l1) - float(torch.nn.functional.cross_entropy(z.detach(), y))) < 1e-6
    # the double-softmax loss is bounded: between -log(e/(e+C-1)) and log(C) + small
    assert 0.55 < float(loss) < 1.2


def test_adam_tf_matches_numpy_oracle_and_differs_from_torch_adam():
    rng = np.random.RandomState(0)
    w = rng.randn(50).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(w.copy()))
    opt = AdamTF([p], lr=1e-3)
    m = np.zeros(50); v = np.zeros(50); wn = w.astype(np.float64)
    for t in range(1, 6):
        g = rng.randn(50).astype(np.float32)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        wn = wn - lr_t * m / (np.sqrt(v) + 1e-8)
    assert np.allclose(p.detach().numpy(), wn, rtol=1e-5, atol=1e-7)
    flat = torch.from_numpy(w.copy()); M = torch.zeros(50); V = torch.zeros(50)
    rng = np.random.RandomState(0); rng.randn(50)
    for t in range(1, 6):
        adam_tf_flat_(flat, torch.from_numpy(rng.randn(50).astype(np.float32)), M, V, t, 1e-3)
    assert np.allclose(flat.numpy(), wn, rtol=1e-5, atol=1e-7)


def test_sgd():
    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.full((4,), 2.0)
    SGD([p], lr=0.01).step()
    assert torch.allclose(p.detach(), torch.full((4,), 0.98))


def test_eager_engine_learns_and_dropout_scaling():
    dd = _dd(600, seed=3)
    eng = EagerMnistEngine(dd, batch_size=50, lr=1e-3, keep_prob=0.7, double_softmax=False, seed=1)
    for _ in range(40):
        eng.train_step()
    loss, acc = eng.read_metrics(0, 40)
    assert loss[-5:].mean() < loss[:5].mean() - 0.3
    l, a = eng.evaluate(dd.images, dd.labels)
    assert a > 0.5
    p, pred = eng.predict(dd.images[:7])
    assert p.shape == (7, 10) and torch.allclose(p.sum(1), torch.ones(7), atol=1e-5)


def test_checkpoint_tf_names_and_exact_resume(tmp_path):
    dd = _dd(300, seed=5)
    a = EagerMnistEngine(dd, batch_size=20, lr=1e-3, keep_prob=1.0, seed=2)
    for _ in range(3):
        a.train_step()
    tensors = ckpt.mnist_state_to_tf(a, with_global_step=True)
    expected = {"Conv1/W_conv1/Variable": (5, 5, 1, 32), "Conv1/b_conv1/Variable": (32,),
                "Conv2/W_conv2/Variable": (5, 5, 32, 64), "Conv2/b_conv2/Variable": (64,),
                "fc1/W_fc1/Variable": (3136, 1024), "fc1/b_fc1/Variable": (1024,),
                "fc2/W_fc2/Variable": (1024, 10), "fc2/b_fc2/Variable": (10,)}
    for name, shape in expected.items():
        assert tensors[name].shape == shape
        assert tensors[name + "/Adam"].shape == shape and tensors[name + "/Adam_1"].shape == shape
    assert int(tensors["Variable"]) == 3
    assert abs(float(tensors["train_step/beta1_power"]) - 0.9 ** 4) < 1e-7
    path = ckpt.Saver(a).save(str(tmp_path / "logs" / "model.ckpt"), global_step=a.global_step)
    assert path.endswith("model.ckpt-3") and ckpt.latest_checkpoint(str(tmp_path / "logs")) == path
    # resume into a fresh engine: the next step must be reproduced exactly
    dd2 = _dd(300, seed=5)
    b = EagerMnistEngine(dd2, batch_size=20, lr=1e-3, keep_prob=1.0, seed=99)
    assert ckpt.Saver(b).restore(path) == 3
    a.train_step(); b.train_step()
    assert torch.equal(a.P, b.P) and torch.equal(a.M, b.M) and b.global_step == 4
    npz = ckpt.export_npz(str(tmp_path / "tf_vars.npz"), tensors)
    z = np.load(npz)
    assert z["fc1__W_fc1__Variable"].shape == (3136, 1024)
