"""``torch.autograd.Function`` wrappers over the HIP kernels (component C16) and the layers built from them.

The production paths (``engine/mnist_engine.py``, ``models/inception_train.py``) run hand-scheduled static
forward/backward programs; these wrappers expose the same kernels to ordinary PyTorch code so that new models can be
composed with ``nn.Module`` and trained with any ``torch.optim`` optimizer (or ``optim.adam_tf.AdamTF``):

* ``linear(x, w, b, relu)``                 — MFMA GEMM with fp32 ``[in, out]`` weights (TensorFlow layout)
* ``conv_bias_relu_pool(x, w, b)``          — stride-1 SAME conv + bias + ReLU + 2x2/2 max-pool in one kernel
* ``conv_bn_relu(x, w, gamma, beta, ...)``  — any Inception geometry, training-mode BatchNorm (statistics in the
                                              conv epilogue) + ReLU, NHWC bf16 in/out
* ``max_pool3x3(x, stride)`` / ``avg_pool3x3_same(x)``

Activations are NHWC bf16, parameters fp32 (converted to bf16 while staged into LDS), gradients of parameters fp32,
gradients of activations bf16 — exactly the conventions of the static engines, whose tests these wrappers share.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import native
from . import hip_ops

BN_EPS = 1e-3
STAT_SLOTS = 16  # DTF_STAT_SLOTS (csrc/hip/gemm_core.h)


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _pad_to(t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    if t.shape[0] == rows and t.shape[1] == cols:
        return t.contiguous()
    out = torch.zeros(rows, cols, dtype=t.dtype, device=t.device)
    out[:t.shape[0], :t.shape[1]] = t
    return out


def _up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class _Linear(torch.autograd.Function):
    # The MFMA kernels need K % 8 == 0 and N % 8 == 0 (N % 2 in the forward).  Shapes that do not fit (the 10-way logits)
    # are zero-padded to the next multiple and run through the same kernels: no library GEMM anywhere on this path.
    @staticmethod
    def forward(ctx, x, w, b, relu: bool):
        K, N = w.shape
        Kp, Np = _up(K, 8), _up(N, 8)
        xb = _pad_to(x, x.shape[0], Kp)
        wp = _pad_to(w, Kp, Np)
        bp = None if b is None else (b if Np == N else torch.cat([b, b.new_zeros(Np - N)]))
        y = hip_ops.fc_forward(xb, wp, bp, relu=relu)[0]
        if Np != N:
            y = y[:, :N].contiguous()
        ctx.save_for_backward(xb, wp, y if relu else None)
        ctx.has_bias, ctx.kn = b is not None, (K, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors  # (padded operands)
        K, N = ctx.kn
        Kp, Np = w.shape
        if y is not None:
            dy = dy * (y > 0)
        dyb = _pad_to(dy.to(torch.bfloat16), dy.shape[0], Np)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = hip_ops.fc_backward_data(dyb, w)
            if Kp != K:
                dx = dx[:, :K].contiguous()
        if ctx.needs_input_grad[1]:
            dw = hip_ops.fc_backward_weight(x, dyb)
            if Kp != K or Np != N:
                dw = dw[:K, :N].contiguous()
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        if dx is not None and x.dtype != torch.float32:
            dx = dx.to(x.dtype)
        return dx, dw, db, None


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """``relu?(x[M,K] @ w[K,N] + b)`` -> fp32; ``x`` bf16 or fp32 (any K, N: odd sizes are zero-padded to multiples of 8)."""
    return _Linear.apply(x, w, b, relu)


class _ConvBiasReluPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        KH, KW = w.shape[0], w.shape[1]
        xb = x.contiguous()
        out, amax = hip_ops.conv_bias_relu_pool(xb, w.reshape(KH * KW * w.shape[2], w.shape[3]).contiguous(), b, KH, KW)
        ctx.save_for_backward(xb, w, out, amax)
        ctx.mark_non_differentiable(amax)
        return out, amax

    @staticmethod
    def backward(ctx, dout, _damax):
        x, w, out, amax = ctx.saved_tensors
        KH, KW, Cin, Cout = w.shape
        B, H, W, _ = x.shape
        dy, db = hip_ops.unpool(dout.float().contiguous(), out, amax, want_db=True)  # through pool + ReLU
        dw = hip_ops.conv_backward_weight(x, dy, KH, KW) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            native.check(native.hip().dtf_conv_bwd_data(dy.data_ptr(), w.contiguous().data_ptr(), dx.data_ptr(), B, H, W, Cin, Cin,
                                                        Cout, KH, KW, 1, (KH - 1) // 2, (KW - 1) // 2, H, W, 0, _st()),
                         "conv_bwd_data")
        return dx, dw, db


def conv_bias_relu_pool(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """bf16 NHWC ``[B,H,W,Cin]`` (Cin % 8 == 0), fp32 HWIO ``w`` -> bf16 ``[B,H/2,W/2,Cout]``."""
    return _ConvBiasReluPool.apply(x, w, b)[0]


class _ConvBNReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, moving_mean, moving_var, stride, pad, momentum):
        lib, chk = native.hip(), native.check
        xb = x.contiguous()
        B, H, W, Cin = xb.shape
        KH, KW, _, Cout = w.shape
        Ho, Wo = (H + 2 * pad[0] - KH) // stride + 1, (W + 2 * pad[1] - KW) // stride + 1
        M = B * Ho * Wo
        dev = xb.device
        z = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
        y = torch.empty_like(z)
        stats = torch.zeros(STAT_SLOTS, 2, Cout, device=dev)
        bn = torch.empty(4, Cout, device=dev)
        wc = w.contiguous()
        chk(lib.dtf_conv_fwd_stats(xb.data_ptr(), wc.data_ptr(), z.data_ptr(), stats.data_ptr(), B, H, W, Cin, Cin, Cout, KH, KW,
                                   stride, pad[0], pad[1], Ho, Wo, _st()), "conv_fwd_stats")
        chk(lib.dtf_bn_finalize(stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), native.ptr(moving_mean),
                                native.ptr(moving_var), bn.data_ptr(), Cout, M, BN_EPS, momentum, _st()), "bn_finalize")
        chk(lib.dtf_bn_relu_fwd(z.data_ptr(), bn.data_ptr(), y.data_ptr(), M, Cout, Cout, _st()), "bn_relu_fwd")
        ctx.save_for_backward(xb, wc, z, bn)
        ctx.geom = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib, chk = native.hip(), native.check
        x, w, z, bn = ctx.saved_tensors
        stride, pad = ctx.geom
        B, H, W, Cin = x.shape
        KH, KW, _, Cout = w.shape
        _, Ho, Wo, _ = z.shape
        M = B * Ho * Wo
        dev = x.device
        dyb = dy.to(torch.bfloat16).contiguous()
        slots = torch.zeros(STAT_SLOTS, 2, Cout, device=dev)
        dgamma, dbeta = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
        dz = z.clone()  # the kernels turn z into dz in place; keep the saved tensor intact for double backward calls
        chk(lib.dtf_bn_relu_bwd_reduce(dyb.data_ptr(), Cout, dz.data_ptr(), bn.data_ptr(), slots.data_ptr(), M, Cout, _st()),
            "bn_bwd_reduce")
        chk(lib.dtf_bn_relu_bwd_apply(dyb.data_ptr(), Cout, dz.data_ptr(), bn.data_ptr(), slots.data_ptr(), 0,
                                      dgamma.data_ptr(), dbeta.data_ptr(), M, Cout, _st()), "bn_bwd_apply")
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            tiles = ((KH * KW * Cin + 127) // 128) * ((Cout + 63) // 64)
            splits = max(1, min(max(1, 1536 // tiles), M // 512 if M >= 512 else 1))
            chk(lib.dtf_conv_dw_general(x.data_ptr(), Cin, dz.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, KH, KW, stride,
                                        pad[0], pad[1], Ho, Wo, splits, _st()), "conv_dw")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            chk(lib.dtf_conv_bwd_data(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cin, Cout, KH, KW, stride, pad[0],
                                      pad[1], Ho, Wo, 0, _st()), "conv_bwd_data")
        return dx, dw, dgamma, dbeta, None, None, None, None, None


def conv_bn_relu(x, w, gamma, beta, moving_mean=None, moving_var=None, stride: int = 1, pad: Tuple[int, int] = (0, 0),
                 momentum: float = 0.9997) -> torch.Tensor:
    """Training-mode ``relu(batchnorm(conv(x, w)))``; bf16 NHWC in/out, fp32 HWIO ``w``; updates the moving statistics
    in place when given."""
    return _ConvBNReLU.apply(x, w, gamma, beta, moving_mean, moving_var, stride, tuple(pad), momentum)


class _MaxPool3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride: int):
        xb = x.contiguous()
        B, H, W, C = xb.shape
        Ho, Wo = (H - 3) // stride + 1, (W - 3) // stride + 1
        out = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=xb.device)
        arg = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=xb.device)
        native.check(native.hip().dtf_maxpool3x3_arg(xb.data_ptr(), out.data_ptr(), arg.data_ptr(), B, H, W, C, C, C, stride, _st()),
                     "maxpool3x3")
        ctx.save_for_backward(arg)
        ctx.shape, ctx.stride = (B, H, W, C), stride
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, H, W, C = ctx.shape
        dyb = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dyb.device)
        native.check(native.hip().dtf_maxpool3x3_bwd(dyb.data_ptr(), C, arg.data_ptr(), dx.data_ptr(), C, B, H, W, C, ctx.stride, 0,
                                                     _st()), "maxpool3x3_bwd")
        return dx, None


class _AvgPool3x3Same(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xb = x.contiguous()
        B, H, W, C = xb.shape
        out = torch.empty_like(xb)
        native.check(native.hip().dtf_avgpool3x3_same(xb.data_ptr(), out.data_ptr(), B, H, W, C, C, C, _st()), "avgpool3x3")
        ctx.shape = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dyb = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty_like(dyb)
        native.check(native.hip().dtf_avgpool3x3_same_bwd(dyb.data_ptr(), C, dx.data_ptr(), C, B, H, W, C, 0, _st()), "avgpool3x3_bwd")
        return dx


def max_pool3x3(x: torch.Tensor, stride: int = 2) -> torch.Tensor:
    return _MaxPool3x3.apply(x, stride)


def avg_pool3x3_same(x: torch.Tensor) -> torch.Tensor:
    return _AvgPool3x3Same.apply(x)


# ------------------------------------------------------------------------------------------------ layers
class Linear(nn.Module):
    """Dense layer with TensorFlow-layout ``[in, out]`` fp32 weights on the MFMA GEMM."""

    def __init__(self, in_features: int, out_features: int, relu: bool = False, std: float = 0.1, bias_init: float = 0.1):
        super().__init__()
        w = torch.empty(in_features, out_features)
        nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)  # truncated_normal(stddev) of demo1/train.py:28-30
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.full((out_features,), bias_init))
        self.relu = relu

    def forward(self, x):
        return linear(x, self.weight, self.bias, self.relu)


class ConvReluPool(nn.Module):
    """``max_pool_2x2(relu(conv2d(x, W) + b))`` of the reference CNN (``demo1/train.py:40-46,67-71``), one kernel."""

    def __init__(self, cin: int, cout: int, k: int = 5, std: float = 0.1, bias_init: float = 0.1):
        super().__init__()
        w = torch.empty(k, k, cin, cout)
        nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.full((cout,), bias_init))

    def forward(self, x):
        return conv_bias_relu_pool(x, self.weight, self.bias)


class ConvBNReLU(nn.Module):
    """Inception-style conv block (no conv bias, BatchNorm eps 1e-3, ReLU), NHWC bf16."""

    def __init__(self, cin: int, cout: int, k=3, stride: int = 1, padding: str = "SAME", momentum: float = 0.9997):
        super().__init__()
        kh, kw = (k, k) if isinstance(k, int) else k
        std = (2.0 / (kh * kw * cin)) ** 0.5
        w = torch.empty(kh, kw, cin, cout)
        nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)
        self.weight = nn.Parameter(w)
        self.gamma = nn.Parameter(torch.ones(cout))
        self.beta = nn.Parameter(torch.zeros(cout))
        self.register_buffer("moving_mean", torch.zeros(cout))
        self.register_buffer("moving_variance", torch.ones(cout))
        self.stride, self.momentum = stride, momentum
        self.pad = ((kh - 1) // 2, (kw - 1) // 2) if padding == "SAME" else (0, 0)

    def forward(self, x):
        if self.training:
            return conv_bn_relu(x, self.weight, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.stride,
                                self.pad, self.momentum)
        scale = self.gamma / torch.sqrt(self.moving_variance + BN_EPS)
        kh, kw, cin, cout = self.weight.shape
        wp = (self.weight * scale).permute(3, 0, 1, 2).reshape(cout, kh * kw * cin).to(torch.bfloat16).contiguous()
        return hip_ops.conv_bias_relu(x.contiguous(), wp, (self.beta - self.moving_mean * scale).contiguous(), Cin=cin, KH=kh,
                                      KW=kw, stride=self.stride, pad=self.pad, relu=True)
