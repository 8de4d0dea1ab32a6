"""autograd Functions / layers over the HIP kernels (ops/autograd.py) against plain PyTorch layers (GPU only)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _r(t):
    return t.to(BF).float()


def test_linear_and_conv_pool_layers_match_torch():
    from distributed_tensorflow_amd.ops import autograd as ag
    torch.manual_seed(0)
    B = 37
    conv = ag.ConvReluPool(32, 64).cuda()
    fc = ag.Linear(7 * 7 * 64, 256, relu=True).cuda()
    out = ag.Linear(256, 10).cuda()
    x = torch.relu(torch.randn(B, 14, 14, 32, device="cuda")).to(BF).requires_grad_(True)
    y = torch.randint(0, 10, (B,), device="cuda")
    h = conv(x)
    logits = out(fc(h.reshape(B, -1)))
    loss = F.cross_entropy(logits, y)
    loss.backward()
    # reference with the same bf16-rounded operands
    xr = x.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wc = _r(conv.weight.detach()).permute(3, 2, 0, 1).requires_grad_(True)
    bc = conv.bias.detach().clone().requires_grad_(True)
    hr = F.max_pool2d(F.relu(F.conv2d(xr, wc, bc, padding=2)), 2)
    hr_b = hr + (_r(hr) - hr).detach()
    w1 = _r(fc.weight.detach()).requires_grad_(True)
    b1 = fc.bias.detach().clone().requires_grad_(True)
    f1 = F.relu(hr_b.permute(0, 2, 3, 1).reshape(B, -1) @ w1 + b1)
    w2 = _r(out.weight.detach()).requires_grad_(True)
    b2 = out.bias.detach().clone().requires_grad_(True)
    lr = F.cross_entropy(f1 @ w2 + b2, y)
    lr.backward()
    assert abs(float(loss) - float(lr)) < 2e-2 * max(1.0, abs(float(lr)))
    assert _l2(h.permute(0, 3, 1, 2), hr) < 1e-2
    assert _l2(out.weight.grad, w2.grad) < 3e-2 and _l2(out.bias.grad, b2.grad) < 3e-2
    assert _l2(fc.weight.grad, w1.grad) < 3e-2 and _l2(fc.bias.grad, b1.grad) < 3e-2
    assert _l2(conv.weight.grad, wc.grad.permute(2, 3, 1, 0)) < 3e-2 and _l2(conv.bias.grad, bc.grad) < 3e-2
    assert _l2(x.grad.permute(0, 3, 1, 2), xr.grad) < 3e-2


def test_conv_bn_relu_block_and_pools_match_torch_and_train():
    from distributed_tensorflow_amd.ops import autograd as ag
    torch.manual_seed(1)
    B = 6

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = ag.ConvBNReLU(8, 32, 3, stride=2, padding="VALID", momentum=0.9)
            self.c2 = ag.ConvBNReLU(32, 48, (1, 7))
            self.c3 = ag.ConvBNReLU(48, 64, (7, 1))
            self.head = ag.Linear(64, 4, std=0.05, bias_init=0.0)

        def forward(self, x):
            h = self.c1(x)
            h = ag.max_pool3x3(h, 2)
            h = self.c3(self.c2(h)) + 0.5 * ag.avg_pool3x3_same(self.c3(self.c2(h)))
            return self.head(h.float().mean((1, 2)))

    net = Net().cuda()
    x = torch.randn(B, 41, 41, 8, device="cuda").to(BF)
    y = torch.tensor([0, 1, 2, 3, 0, 1], device="cuda")
    opt = torch.optim.SGD(net.parameters(), lr=0.2)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = F.cross_entropy(net(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.75 * losses[0] and all(l == l for l in losses), losses
    assert float(net.c1.moving_mean.abs().sum()) > 0  # moving statistics updated in training mode
    net.eval()
    with torch.no_grad():
        ev = net(x)
    assert ev.shape == (B, 4) and torch.isfinite(ev).all()
    # one block against autograd
    blk = ag.ConvBNReLU(16, 32, 3).cuda()
    xi = torch.randn(4, 9, 9, 16, device="cuda").to(BF).requires_grad_(True)
    g = torch.randn(4, 9, 9, 32, device="cuda").to(BF)
    yo = blk(xi)
    yo.backward(g)
    xr = xi.detach().float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = _r(blk.weight.detach()).permute(3, 2, 0, 1).requires_grad_(True)
    z = F.conv2d(xr, wr, padding=1)
    z = z + (_r(z) - z).detach()
    yr = F.relu(F.batch_norm(z, None, None, blk.gamma.detach(), blk.beta.detach(), True, 0.0, 1e-3))
    yr.backward(g.float().permute(0, 3, 1, 2))
    assert _l2(yo.permute(0, 3, 1, 2), yr) < 1e-2
    assert _l2(blk.weight.grad, wr.grad.permute(2, 3, 1, 0)) < 2e-2 and _l2(xi.grad.permute(0, 3, 1, 2), xr.grad) < 2e-2
