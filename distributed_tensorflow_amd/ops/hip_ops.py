"""Thin tensor-level wrappers over the HIP kernels in ``libdtf_hip.so``.

Every function takes/returns torch tensors on the GPU, validates dtypes/contiguity on the
host (the kernels assume them) and launches on the *current* torch stream, so the ops
compose with torch code and can be captured into a hipGraph.  These are the building
blocks used by the autograd Functions in ``ops.autograd`` and by the kernel unit tests;
the MNIST training hot path calls the same kernels from C++ (``csrc/hip/engine.hip``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .. import native


def _st() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA tensor of %s (got %s, cuda=%s, contiguous=%s)"
                         % (name, dtype, t.dtype, t.is_cuda, t.is_contiguous()))


def fc_forward(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
               splits: int = 1) -> torch.Tensor:
    """``x[M,K] (bf16|fp32) @ w[K,N] (fp32)`` -> fp32 ``[S,M,N]`` split-K slabs (S==1: bias/relu applied)."""
    assert x.dtype in (torch.bfloat16, torch.float32)
    _chk(x, x.dtype, "x")
    _chk(w, torch.float32, "w")
    M, K = x.shape
    K2, N = w.shape
    assert K == K2
    ksteps = (K + 63) // 64  # BK of the fc_fwd kernel (gemm_api.hip)
    kchunk = (ksteps + splits - 1) // splits * 64
    s_eff = (K + kchunk - 1) // kchunk
    out = torch.empty(s_eff, M, N, dtype=torch.float32, device=x.device)
    native.check(native.hip().dtf_fc_fwd(x.data_ptr(), int(x.dtype == torch.float32), w.data_ptr(), out.data_ptr(),
                                         M, N, K, splits, native.ptr(bias), int(relu), _st()), "fc_fwd")
    return out


def fc_backward_data(dy: torch.Tensor, w: torch.Tensor, splits: int = 1) -> torch.Tensor:
    """``dy[M,K] (bf16) @ w[N,K]^T (fp32)`` -> fp32 ``[M,N]``."""
    _chk(dy, torch.bfloat16, "dy")
    _chk(w, torch.float32, "w")
    M, K = dy.shape
    N, K2 = w.shape
    assert K == K2
    out = torch.zeros(M, N, dtype=torch.float32, device=dy.device)
    native.check(native.hip().dtf_fc_bwd_data(dy.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, splits, _st()),
                 "fc_bwd_data")
    return out


def fc_backward_weight(x: torch.Tensor, dy: torch.Tensor, splits: int = 1, out: Optional[torch.Tensor] = None
                       ) -> torch.Tensor:
    """``x[Kb,M]^T (bf16|fp32) @ dy[Kb,N] (bf16)`` -> fp32 ``[M,N]`` (Kb = batch)."""
    assert x.dtype in (torch.bfloat16, torch.float32)
    _chk(x, x.dtype, "x")
    _chk(dy, torch.bfloat16, "dy")
    Kb, M = x.shape
    Kb2, N = dy.shape
    assert Kb == Kb2
    if out is None:
        out = torch.zeros(M, N, dtype=torch.float32, device=x.device)
    native.check(native.hip().dtf_fc_dw(x.data_ptr(), int(x.dtype == torch.float32), dy.data_ptr(), out.data_ptr(),
                                        M, N, Kb, splits, _st()), "fc_dw")
    return out


def conv1_forward(images: torch.Tensor, w: torch.Tensor, b: torch.Tensor, index: Optional[torch.Tensor] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv5x5(1->32)+bias+ReLU+pool: fp32 ``[N,784]`` (gathered by ``index``) -> bf16 ``[B,14,14,32]``, argmax."""
    _chk(images, torch.float32, "images")
    _chk(w, torch.float32, "w")
    B = images.shape[0] if index is None else index.shape[0]
    out = torch.empty(B, 14, 14, 32, dtype=torch.bfloat16, device=images.device)
    amax = torch.empty(B, 14, 14, 32, dtype=torch.uint8, device=images.device)
    native.check(native.hip().dtf_conv1_fwd(images.data_ptr(), native.ptr(index), 0, 1, B, w.data_ptr(), b.data_ptr(),
                                            out.data_ptr(), amax.data_ptr(), _st()), "conv1_fwd")
    return out, amax


def conv1_backward_weight(images: torch.Tensor, dpool: torch.Tensor, amax: torch.Tensor,
                          index: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(images, torch.float32, "images")
    _chk(dpool, torch.float32, "dpool")
    _chk(amax, torch.uint8, "amax")
    B = dpool.shape[0]
    dW = torch.zeros(25, 32, dtype=torch.float32, device=images.device)
    db = torch.zeros(32, dtype=torch.float32, device=images.device)
    native.check(native.hip().dtf_conv1_dw(images.data_ptr(), native.ptr(index), 0, 1, B, dpool.data_ptr(),
                                           amax.data_ptr(), dW.data_ptr(), db.data_ptr(), _st()), "conv1_dw")
    return dW, db


def conv_bias_relu_pool(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, KH: int, KW: int
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 NHWC ``[B,H,W,Cin]``, fp32 HWIO weights -> bf16 ``[B,H/2,W/2,Cout]`` + uint8 argmax."""
    _chk(x, torch.bfloat16, "x")
    _chk(w, torch.float32, "w")
    B, H, W, Cin = x.shape
    Cout = w.shape[-1]
    out = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.bfloat16, device=x.device)
    amax = torch.empty(B, H // 2, W // 2, Cout, dtype=torch.uint8, device=x.device)
    native.check(native.hip().dtf_conv_bias_relu_pool(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                      amax.data_ptr(), B, H, W, Cin, Cout, KH, KW, _st()), "conv_pool")
    return out, amax


def unpool(dpool: torch.Tensor, pooled: torch.Tensor, amax: torch.Tensor, want_db: bool = True
           ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """max-pool(2x2)+ReLU backward: fp32 ``[B,Hp,Wp,C]`` -> dense bf16 ``[B,2Hp,2Wp,C]`` (+ bias grad)."""
    _chk(dpool, torch.float32, "dpool")
    _chk(pooled, torch.bfloat16, "pooled")
    _chk(amax, torch.uint8, "amax")
    B, Hp, Wp, Cc = dpool.shape
    dy = torch.empty(B, 2 * Hp, 2 * Wp, Cc, dtype=torch.bfloat16, device=dpool.device)
    db = torch.zeros(Cc, dtype=torch.float32, device=dpool.device) if want_db else None
    native.check(native.hip().dtf_unpool(dpool.data_ptr(), pooled.data_ptr(), amax.data_ptr(), dy.data_ptr(),
                                         native.ptr(db), B, Hp, Wp, Cc, _st()), "unpool")
    return dy, db


def conv_backward_data_mask(dy: torch.Tensor, w: torch.Tensor, mask: torch.Tensor, KH: int, KW: int) -> torch.Tensor:
    _chk(dy, torch.bfloat16, "dy")
    _chk(w, torch.float32, "w")
    _chk(mask, torch.bfloat16, "mask")
    B, H, W, Cout = dy.shape
    Cin = mask.shape[-1]
    dx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device)
    native.check(native.hip().dtf_conv_bwd_data_mask(dy.data_ptr(), w.data_ptr(), mask.data_ptr(), dx.data_ptr(),
                                                     B, H, W, Cin, Cout, KH, KW, _st()), "conv_bwd_data")
    return dx


def conv_backward_weight(x: torch.Tensor, dy: torch.Tensor, KH: int, KW: int, splits: int = 32) -> torch.Tensor:
    _chk(x, torch.bfloat16, "x")
    _chk(dy, torch.bfloat16, "dy")
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    dW = torch.zeros(KH, KW, Cin, Cout, dtype=torch.float32, device=x.device)
    native.check(native.hip().dtf_conv_dw(x.data_ptr(), dy.data_ptr(), dW.data_ptr(), B, H, W, Cin, Cout, KH, KW,
                                          splits, _st()), "conv_dw")
    return dW


def conv2_backward_lds(dpool2: torch.Tensor, pool2: torch.Tensor, amax2: torch.Tensor, w: torch.Tensor,
                       pool1: torch.Tensor, images_per_block: int = 4):
    """Fused MNIST conv2 backward (un-pool + bwd-data + ReLU mask + db, and dW) -> (dpool1, dW, db2)."""
    _chk(dpool2, torch.float32, "dpool2")
    _chk(pool2, torch.bfloat16, "pool2")
    _chk(amax2, torch.uint8, "amax2")
    _chk(w, torch.float32, "w")
    _chk(pool1, torch.bfloat16, "pool1")
    B = dpool2.shape[0]
    assert dpool2.shape[1:] == (7, 7, 64) and pool1.shape == (B, 14, 14, 32) and w.numel() == 25 * 32 * 64
    dpool1 = torch.empty(B, 14, 14, 32, dtype=torch.float32, device=dpool2.device)
    dW = torch.zeros(5, 5, 32, 64, dtype=torch.float32, device=dpool2.device)
    db = torch.zeros(64, dtype=torch.float32, device=dpool2.device)
    native.check(native.hip().dtf_conv2_bwd_lds(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), w.data_ptr(),
                                                pool1.data_ptr(), dpool1.data_ptr(), dW.data_ptr(), db.data_ptr(), B,
                                                images_per_block, _st()), "conv2_bwd_lds")
    return dpool1, dW, db


def conv_bias_relu(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, *, Cin: int, KH: int, KW: int,
                   stride: int = 1, pad: Tuple[int, int] = (0, 0), relu: bool = True,
                   out: Optional[torch.Tensor] = None, out_channel_offset: int = 0,
                   in_channel_offset: int = 0, core: int = 1) -> torch.Tensor:
    """Inference conv (+folded BN bias, ReLU): bf16 NHWC in (optionally a channel slice), bf16 packed
    ``[Cout, KH*KW*Cin]`` weights, bf16 NHWC out written into ``out[..., off:off+Cout]`` (concat by slice write).
    ``core``: 1 = register-staged GEMM core (gemm_core.h), 2..5 = async-LDS core variants (igemm2.hip;
    needs ``igemm2_init()``)."""
    _chk(x, torch.bfloat16, "x")
    _chk(w_packed, torch.bfloat16, "w_packed")
    _chk(bias, torch.float32, "bias")
    B, H, W, Ctot = x.shape
    Cout, K = w_packed.shape
    assert K == KH * KW * Cin and in_channel_offset + Cin <= Ctot
    Ho = (H + 2 * pad[0] - KH) // stride + 1
    Wo = (W + 2 * pad[1] - KW) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=x.device)
    _chk(out, torch.bfloat16, "out")
    assert out.shape[:3] == (B, Ho, Wo) and out_channel_offset + Cout <= out.shape[3]
    xp = x.data_ptr() + 2 * in_channel_offset
    op = out.data_ptr() + 2 * out_channel_offset
    args = (xp, w_packed.data_ptr(), bias.data_ptr(), op, B, H, W, Cin, Ctot, Cout, out.shape[3], KH, KW, stride, pad[0], pad[1],
            Ho, Wo, int(relu))
    if core >= 2:  # async-LDS core, variant = core - 2 (bit 0: 128-row tiles, bit 1: 64-column tiles)
        rc = native.hip().dtf_conv_bias_relu_v2(*args, core - 2, _st())
    else:
        rc = native.hip().dtf_conv_bias_relu(*args, _st())
    native.check(rc, "conv_bias_relu(core %d)" % core)
    return out


def igemm2_init() -> None:
    """One-time setup of the async-LDS conv core (raises its dynamic-LDS limits; not capturable)."""
    native.check(native.hip().dtf_igemm2_init() or native.hip().dtf_igemm2_dw_init(), "igemm2_init")


def adam_tf_(P: torch.Tensor, G: torch.Tensor, M: torch.Tensor, V: torch.Tensor, step: torch.Tensor,
             ticket: torch.Tensor, *, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
             grad_scale: float = 1.0, zero_from: int = 0) -> None:
    op = native.OptimParams()
    op.P, op.G, op.M, op.V = P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr()
    op.n, op.zero_from = P.numel(), zero_from
    op.lr, op.beta1, op.beta2, op.eps, op.grad_scale = lr, beta1, beta2, eps, grad_scale
    op.step, op.ticket, op.metrics_ring, op.ring = step.data_ptr(), ticket.data_ptr(), 0, 0
    native.check(native.hip().dtf_adam_tf(C.byref(op), _st()), "adam_tf")


def sgd_(P: torch.Tensor, G: torch.Tensor, step: torch.Tensor, ticket: torch.Tensor, *, lr: float,
         grad_scale: float = 1.0, zero_from: int = 0) -> None:
    op = native.OptimParams()
    op.P, op.G, op.M, op.V = P.data_ptr(), G.data_ptr(), 0, 0
    op.n, op.zero_from = P.numel(), zero_from
    op.lr, op.grad_scale = lr, grad_scale
    op.step, op.ticket, op.metrics_ring, op.ring = step.data_ptr(), ticket.data_ptr(), 0, 0
    native.check(native.hip().dtf_sgd(C.byref(op), _st()), "sgd")


# ----------------------------------------------------------------------------- exact-fp32 (reference precision) ops
# csrc/hip/f32_fc.hip, f32_conv2.hip: v_mfma_f32_16x16x4_f32, fp32 operands and accumulation (demo1/train.py computes in
# float32); what NativeMnistEngine(dtype="fp32") runs.

def fc_forward_f32(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                   splits: int = 1) -> torch.Tensor:
    """``x[M,K] @ w[K,N]`` (both fp32) -> fp32 ``[S,M,N]`` split-K slabs (S==1: bias/relu applied)."""
    _chk(x, torch.float32, "x")
    _chk(w, torch.float32, "w")
    M, K = x.shape
    K2, N = w.shape
    assert K == K2
    ksteps = (K + 31) // 32  # BK of the kernel (f32_fc.hip)
    kchunk = (ksteps + splits - 1) // splits * 32
    s_eff = (K + kchunk - 1) // kchunk
    out = torch.empty(s_eff, M, N, dtype=torch.float32, device=x.device)
    native.check(native.hip().dtf_fc_fwd_f32(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, splits, native.ptr(bias),
                                             int(relu), _st()), "fc_fwd_f32")
    return out


def fc_backward_data_f32(dy: torch.Tensor, w: torch.Tensor, splits: int = 1) -> torch.Tensor:
    """``dy[M,K] @ w[N,K]^T`` (fp32) -> fp32 ``[M,N]``."""
    _chk(dy, torch.float32, "dy")
    _chk(w, torch.float32, "w")
    M, K = dy.shape
    N, K2 = w.shape
    assert K == K2
    out = torch.zeros(M, N, dtype=torch.float32, device=dy.device)
    native.check(native.hip().dtf_fc_bwd_data_f32(dy.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, splits, _st()),
                 "fc_bwd_data_f32")
    return out


def fc_backward_weight_f32(x: torch.Tensor, dy: torch.Tensor, n_valid: Optional[int] = None, splits: int = 1) -> torch.Tensor:
    """``x[Kb,M]^T @ dy[Kb,N]`` (fp32, Kb = batch) -> fp32 ``[M,N]``.  ``n_valid``: use only the first columns of ``dy``
    (rows padded to a multiple of 4 floats, e.g. the [B,16] logit-gradient buffer holding 10 classes)."""
    _chk(x, torch.float32, "x")
    _chk(dy, torch.float32, "dy")
    Kb, M = x.shape
    Kb2, ldy = dy.shape
    N = ldy if n_valid is None else n_valid
    assert Kb == Kb2 and N <= ldy
    out = torch.zeros(M, N, dtype=torch.float32, device=x.device)
    native.check(native.hip().dtf_fc_dw_f32(x.data_ptr(), M, dy.data_ptr(), ldy, out.data_ptr(), M, N, Kb, splits, _st()),
                 "fc_dw_f32")
    return out


def conv1_forward_f32(images: torch.Tensor, w: torch.Tensor, b: torch.Tensor, index: Optional[torch.Tensor] = None
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv5x5(1->32)+bias+ReLU+pool with fp32 output ``[B,14,14,32]`` and the pool argmax."""
    _chk(images, torch.float32, "images")
    _chk(w, torch.float32, "w")
    B = images.shape[0] if index is None else index.shape[0]
    out = torch.empty(B, 14, 14, 32, dtype=torch.float32, device=images.device)
    amax = torch.empty(B, 14, 14, 32, dtype=torch.uint8, device=images.device)
    native.check(native.hip().dtf_conv1_fwd_f32(images.data_ptr(), native.ptr(index), 0, 1, B, w.data_ptr(), b.data_ptr(),
                                                out.data_ptr(), amax.data_ptr(), _st()), "conv1_fwd_f32")
    return out, amax


def conv2_forward_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """MNIST conv2 (5x5, 32->64, SAME) + bias + ReLU + 2x2 pool: fp32 ``[B,14,14,32]`` -> fp32 ``[B,7,7,64]`` + argmax."""
    _chk(x, torch.float32, "x")
    _chk(w, torch.float32, "w")
    _chk(b, torch.float32, "b")
    B = x.shape[0]
    assert x.shape[1:] == (14, 14, 32) and w.numel() == 25 * 32 * 64 and b.numel() == 64
    out = torch.empty(B, 7, 7, 64, dtype=torch.float32, device=x.device)
    amax = torch.empty(B, 7, 7, 64, dtype=torch.uint8, device=x.device)
    native.check(native.hip().dtf_conv2_fwd_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), amax.data_ptr(), B,
                                                _st()), "conv2_fwd_f32")
    return out, amax


def conv2_backward_f32(dpool2: torch.Tensor, pool2: torch.Tensor, amax2: torch.Tensor, w: torch.Tensor, pool1: torch.Tensor):
    """MNIST conv2 backward at fp32 (un-pool + bwd-data + ReLU mask + db, and dW) -> (dpool1, dW, db2)."""
    _chk(dpool2, torch.float32, "dpool2")
    _chk(pool2, torch.float32, "pool2")
    _chk(amax2, torch.uint8, "amax2")
    _chk(w, torch.float32, "w")
    _chk(pool1, torch.float32, "pool1")
    B = dpool2.shape[0]
    assert dpool2.shape[1:] == (7, 7, 64) and pool1.shape == (B, 14, 14, 32) and w.numel() == 25 * 32 * 64
    dpool1 = torch.empty(B, 14, 14, 32, dtype=torch.float32, device=dpool2.device)
    dW = torch.zeros(5, 5, 32, 64, dtype=torch.float32, device=dpool2.device)
    db = torch.zeros(64, dtype=torch.float32, device=dpool2.device)
    lib = native.hip()
    native.check(lib.dtf_conv2_bwd_data_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), w.data_ptr(),
                                            pool1.data_ptr(), dpool1.data_ptr(), B, None, _st()), "conv2_bwd_data_f32")
    native.check(lib.dtf_conv_dw_duo_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), pool1.data_ptr(), dW.data_ptr(),
                                         db.data_ptr(), B, 0, 0, 0, 1, 0, 0, 0, 0, None, _st()), "conv2_dw_f32")
    return dpool1, dW, db
