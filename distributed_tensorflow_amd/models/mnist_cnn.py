"""MNIST CNN (component C3) — parameter layout, initialisers and the PyTorch oracle.

Architecture (``demo1/train.py:49-123``): 784 -> [B,28,28,1] -> conv5x5(1->32, SAME)+b
-> ReLU -> maxpool2x2 -> conv5x5(32->64, SAME)+b -> ReLU -> maxpool2x2 -> flatten 3136
(NHWC order h,w,c) -> FC 1024 + ReLU -> dropout(keep_prob) -> FC 10 -> softmax.
Loss = mean softmax-cross-entropy applied to the *already softmaxed* prediction
(``demo1/train.py:123,127`` — the double softmax, quirk Q1, default on).

Parameters are kept in the reference's TensorFlow layouts (conv filters HWIO, dense
[in,out]) both here and in the native engine's flat arena, so checkpoints export to
the reference variable names without any permutation.

This module is the *correctness oracle* and the eager "measured baseline"; the
shipped hot path is ``engine.mnist_engine.NativeMnistEngine`` (HIP kernels).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

# name -> (TF variable name, shape, kind)
PARAM_SPECS = OrderedDict([
    ("W_fc1", ("fc1/W_fc1/Variable", (3136, 1024), "weight")),
    ("b_fc1", ("fc1/b_fc1/Variable", (1024,), "bias")),
    ("W_fc2", ("fc2/W_fc2/Variable", (1024, 10), "weight")),
    ("b_fc2", ("fc2/b_fc2/Variable", (10,), "bias")),
    ("W_conv2", ("Conv2/W_conv2/Variable", (5, 5, 32, 64), "weight")),
    ("b_conv2", ("Conv2/b_conv2/Variable", (64,), "bias")),
    ("W_conv1", ("Conv1/W_conv1/Variable", (5, 5, 1, 32), "weight")),
    ("b_conv1", ("Conv1/b_conv1/Variable", (32,), "bias")),
])
# ^ arena order == order in which gradients become available in backward (fc2/fc1 first,
#   conv1 last) so the data-parallel reducer can all-reduce contiguous buckets early.

NUM_PARAMS = 3274634  # 800+32+51200+64+3211264+1024+10240+10


def numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


def truncated_normal_(t: torch.Tensor, stddev: float, generator=None) -> torch.Tensor:
    """``tf.truncated_normal``: resample beyond 2 sigma (C17)."""
    return torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev,
                                       generator=generator)


def init_params(seed: Optional[int] = None, device="cpu", dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """``weight_variable`` = truncated_normal(stddev=0.1), ``bias_variable`` = 0.1
    (``demo1/train.py:28-36``)."""
    gen = None
    if seed is not None:
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, (_tf, shape, kind) in PARAM_SPECS.items():
        t = torch.empty(shape, dtype=torch.float32)
        if kind == "weight":
            truncated_normal_(t, 0.1, gen)
        else:
            t.fill_(0.1)
        out[name] = t.to(device=device, dtype=dtype)
    return out


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def forward_logits(params: Dict[str, torch.Tensor], x: torch.Tensor, *, keep_prob: float = 1.0,
                   dropout_mask: Optional[torch.Tensor] = None, training: bool = False,
                   bf16_sim: bool = False, return_intermediates: bool = False,
                   fc1_gate: Optional[torch.Tensor] = None):
    """Plain-PyTorch forward.  ``x`` [B,784] fp32.  Returns logits [B,10].

    ``bf16_sim`` rounds the operands of conv2 / fc1 (and the activations the native
    path stores as bf16) so kernel tests can use tight tolerances.
    ``dropout_mask`` [B,1024] of {0,1}: explicit mask (tests); else Bernoulli(keep_prob).
    ``fc1_gate`` [B,1024] of {0,1}: the combined ReLU-and-dropout gate of another run (tests): fc1 becomes
    ``pre * gate / keep_prob``, so a pre-activation whose sign is decided by rounding noise cannot flip the
    ReLU mask between the two implementations (one such element on a high-gradient row moves a whole
    column of dW_fc1 by >10 %).
    """
    r = _bf16_round if bf16_sim else (lambda t: t)
    B = x.shape[0]
    xi = x.reshape(B, 1, 28, 28)
    w1 = params["W_conv1"].permute(3, 2, 0, 1)  # HWIO -> OIHW
    c1 = F.conv2d(xi, w1, params["b_conv1"], stride=1, padding=2)
    p1 = r(F.max_pool2d(F.relu(c1), 2))
    w2 = r(params["W_conv2"]).permute(3, 2, 0, 1)
    c2 = F.conv2d(p1, w2, params["b_conv2"], stride=1, padding=2)
    p2 = r(F.max_pool2d(F.relu(c2), 2))
    flat = p2.permute(0, 2, 3, 1).reshape(B, 3136)  # NHWC flatten (K7)
    pre = flat @ r(params["W_fc1"]) + params["b_fc1"]
    h = F.relu(pre) if fc1_gate is None else pre * fc1_gate / keep_prob
    if fc1_gate is None and training and keep_prob < 1.0:
        if dropout_mask is None:
            dropout_mask = (torch.rand_like(h) < keep_prob).to(h.dtype)
        h = h * dropout_mask / keep_prob  # tf.nn.dropout scaling (C13)
    hd = r(h)
    logits = hd @ params["W_fc2"] + params["b_fc2"]
    if return_intermediates:
        return logits, {"pool1": p1, "pool2": p2, "flat": flat, "h_drop": hd}
    return logits


def softmax_xent(logits: torch.Tensor, labels: torch.Tensor, *, double_softmax: bool = True
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(mean loss, prediction probs).  ``labels`` int64 [B] or one-hot [B,C]."""
    p = torch.softmax(logits, dim=1)
    z = p if double_softmax else logits
    logq = torch.log_softmax(z, dim=1)
    if labels.dim() == 2:
        loss = -(labels * logq).sum(1).mean()
    else:
        loss = F.nll_loss(logq, labels.long())
    return loss, p


def accuracy(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """``mean(argmax(pred)==argmax(y))`` (C5, ``demo1/train.py:135-141``)."""
    y = labels.argmax(1) if labels.dim() == 2 else labels.long()
    return (pred.argmax(1) == y).float().mean()


class MnistCNN(torch.nn.Module):
    """Eager PyTorch statement of the model (oracle / baseline).  Parameters are
    ``nn.Parameter``s in TF layout, registered in arena order."""

    def __init__(self, seed: Optional[int] = None, double_softmax: bool = True):
        super().__init__()
        self.double_softmax = double_softmax
        for k, v in init_params(seed).items():
            self.register_parameter(k, torch.nn.Parameter(v))

    def param_dict(self) -> Dict[str, torch.Tensor]:
        return OrderedDict((k, getattr(self, k)) for k in PARAM_SPECS)

    def forward(self, x, keep_prob: float = 1.0, dropout_mask=None):
        return forward_logits(self.param_dict(), x, keep_prob=keep_prob, dropout_mask=dropout_mask,
                              training=self.training)

    def loss(self, x, labels, keep_prob: float = 1.0, dropout_mask=None):
        logits = self.forward(x, keep_prob, dropout_mask)
        loss, pred = softmax_xent(logits, labels, double_softmax=self.double_softmax)
        return loss, pred
