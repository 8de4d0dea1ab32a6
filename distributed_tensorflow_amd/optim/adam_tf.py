"""Adam with TensorFlow-1.x semantics (component C6) and plain SGD (C7).

``tf.train.AdamOptimizer(1e-4)`` (``demo1/train.py:132``) applies

    lr_t  = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    m     = beta1*m + (1-beta1)*g ;  v = beta2*v + (1-beta2)*g*g
    theta = theta - lr_t * m / (sqrt(v) + eps)

i.e. epsilon sits *outside* the bias correction, unlike ``torch.optim.Adam``.  The
native engine runs the same update as one multi-tensor HIP launch over a flat arena
(``csrc/hip/optim.hip``); the classes here are the eager/oracle implementation and
the CPU fallback, operating either on parameter lists or on one flat tensor.
"""
from __future__ import annotations

import math
from typing import Iterable, List

import torch


class AdamTF(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, beta1: float = 0.9,
                 beta2: float = 0.999, eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2, eps, lr = group["beta1"], group["beta2"], group["eps"], group["lr"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p)
                    st["v"] = torch.zeros_like(p)
                st["step"] += 1
                t = st["step"]
                lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
                st["m"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st["v"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                p.addcdiv_(st["m"], st["v"].sqrt().add_(eps), value=-lr_t)
        return loss


def adam_tf_flat_(param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                  step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999,
                  eps: float = 1e-8, grad_scale: float = 1.0) -> None:
    """In-place TF-Adam on flat tensors (``step`` is the 1-based update index)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    g = grad * grad_scale if grad_scale != 1.0 else grad
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    param.addcdiv_(m, v.sqrt().add_(eps), value=-lr_t)


class SGD(torch.optim.Optimizer):
    """``GradientDescentOptimizer(lr)`` (``retrain1/retrain.py:287``)."""

    def __init__(self, params: Iterable[torch.Tensor], lr: float = 0.01):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.add_(p.grad, alpha=-group["lr"])
        return loss
