"""The retrained softmax layer (component C36) and the backbones that feed it.

``add_final_training_ops`` (``retrain1/retrain.py:262-297``): ``final_weights`` [2048,C]
truncated-normal sigma=0.001, ``final_biases`` zeros, ``logits = x W + b``,
``final_result = softmax(logits)``, cross-entropy taken on the already-softmaxed tensor (double
softmax, Q1), plain SGD with a global step, accuracy = mean(argmax == label).

``RetrainHead`` runs the step natively (``csrc/hip/retrain_head.hip``: gather-by-row + GEMV + softmax
+ loss + dlogits in one launch, dW/db + fused SGD in a second) or with PyTorch ops on CPU.  Parameters
and gradients live in one flat tensor so the data-parallel path all-reduces a single 41 KB message.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import native
from .mnist_cnn import truncated_normal_

BOTTLENECK = 2048


class RetrainHeadParams(C.Structure):  # csrc/hip/retrain_head.hip
    _fields_ = [("cache", C.c_void_p), ("labels_all", C.c_void_p), ("rows", C.c_void_p), ("W", C.c_void_p),
                ("b", C.c_void_p), ("B", C.c_int), ("C", C.c_int), ("grad_scale", C.c_float),
                ("double_softmax", C.c_int), ("training", C.c_int), ("probs", C.c_void_p), ("dz", C.c_void_p),
                ("pred", C.c_void_p), ("metrics", C.c_void_p), ("logits", C.c_void_p)]


class RetrainHead:
    def __init__(self, class_count: int, device, *, learning_rate: float = 0.01, double_softmax: bool = True,
                 seed: Optional[int] = None, final_tensor_name: str = "final_result"):
        self.C = class_count
        self.device = torch.device(device)
        if self.device.type == "cuda":
            n = native.hip().dtf_sizeof_retrain_head_params()
            if n != C.sizeof(RetrainHeadParams):
                raise native.NativeError("RetrainHeadParams mirror is %d bytes, the library's struct %d" % (C.sizeof(RetrainHeadParams), n))
        self.lr = learning_rate
        self.double_softmax = double_softmax
        self.final_tensor_name = final_tensor_name
        self.native = self.device.type == "cuda"
        n = BOTTLENECK * class_count + class_count
        self.n_pad = (n + 3) // 4 * 4
        self.P = torch.zeros(self.n_pad, dtype=torch.float32, device=self.device)
        self.G = torch.zeros(self.n_pad, dtype=torch.float32, device=self.device)
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(seed)
        w = truncated_normal_(torch.empty(BOTTLENECK, class_count), 0.001, gen)
        self.weights.copy_(w.to(self.device))
        self.global_step = 0
        self.world = 1
        self._metrics = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._step_t = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._dz = None
        # data parallelism (retrain2): the framework's own communicator (parallel/rccl_comm.py) when one is attached — the
        # 41 KB gradient all-reduce is then an enqueue on the step's stream, no torch.distributed call and no host sync in
        # the step; None = the process group (gloo on CPU / several ranks sharing one GPU)
        self.comm = None
        self._ring = None  # optional [R,4] device ring of per-step metric sums (train ce, train correct, val ce, val correct)

    def use_metrics_ring(self, ring: Optional[torch.Tensor]) -> None:
        """Per-step summaries without a host sync per step: ``train_step(..., ring_slot=s)`` adds the batch's loss sum and
        correct count to ``ring[s, 0:2]``, ``accumulate_metrics(..., ring_slot=s, column=2)`` a validation batch's to
        ``ring[s, 2:4]``; the driver reads and zeroes the ring at logging points."""
        assert ring is None or (ring.dim() == 2 and ring.shape[1] == 4 and ring.dtype == torch.float32 and ring.is_contiguous())
        self._ring = ring

    # flat views: TF names final_training_ops/weights/final_weights, .../biases/final_biases
    @property
    def weights(self) -> torch.Tensor:
        return self.P[:BOTTLENECK * self.C].view(BOTTLENECK, self.C)

    @property
    def biases(self) -> torch.Tensor:
        return self.P[BOTTLENECK * self.C:BOTTLENECK * self.C + self.C]

    def tf_tensors(self) -> Dict[str, np.ndarray]:
        return {"final_training_ops/weights/final_weights": self.weights.detach().cpu().numpy().copy(),
                "final_training_ops/biases/final_biases": self.biases.detach().cpu().numpy().copy(),
                "Variable": np.asarray(self.global_step, np.int32)}

    def load_tf_tensors(self, t: Dict[str, np.ndarray]) -> None:
        self.weights.copy_(torch.from_numpy(np.asarray(t["final_training_ops/weights/final_weights"], np.float32)).to(self.device))
        self.biases.copy_(torch.from_numpy(np.asarray(t["final_training_ops/biases/final_biases"], np.float32)).to(self.device))
        self.global_step = int(np.asarray(t.get("Variable", 0)).reshape(-1)[0])

    # ------------------------------------------------------------------ forward / eval
    def _fwd_native(self, cache, labels_all, rows, training: bool, want_probs: bool, metrics_ptr: int = 0, capture=None):
        """``capture``: optional (probs [B,C], logits [B,C]) fp32 device tensors the kernel also fills (the activations /
        pre_activations the reference histograms on every step)."""
        B = int(rows.shape[0]) if rows is not None else int(cache.shape[0])
        p = RetrainHeadParams()
        p.cache, p.labels_all, p.rows = cache.data_ptr(), labels_all.data_ptr(), native.ptr(rows)
        p.W, p.b = self.P.data_ptr(), self.P.data_ptr() + 4 * BOTTLENECK * self.C  # (= weights / biases: views cost host time)
        p.B, p.C, p.grad_scale = B, self.C, 1.0 / B
        p.double_softmax, p.training = int(self.double_softmax), int(training)
        probs = torch.empty(B, self.C, dtype=torch.float32, device=self.device) if want_probs else None
        pred = torch.empty(B, dtype=torch.int32, device=self.device) if want_probs else None
        if training and (self._dz is None or self._dz.shape[0] != B):
            self._dz = torch.empty(B, self.C, dtype=torch.float32, device=self.device)
        p.probs, p.pred, p.dz = native.ptr(probs), native.ptr(pred), native.ptr(self._dz if training else None)
        if capture is not None:
            assert capture[0].shape == (B, self.C) and capture[1].shape == (B, self.C)
            p.probs, p.logits = capture[0].data_ptr(), capture[1].data_ptr()
        if metrics_ptr:
            p.metrics = metrics_ptr  # (a slot of the per-step ring: accumulates, zeroed by the driver)
        else:
            self._metrics.zero_()
            p.metrics = self._metrics.data_ptr()
        native.check(native.hip().dtf_retrain_head_fwd(C.byref(p), torch.cuda.current_stream().cuda_stream), "head fwd")
        return B, probs, pred

    def _fwd_eager(self, x, y):
        logits = x @ self.weights + self.biases
        p = torch.softmax(logits, 1)
        z = p if self.double_softmax else logits
        loss = torch.nn.functional.nll_loss(torch.log_softmax(z, 1), y.long())
        return loss, p

    def evaluate(self, cache: torch.Tensor, labels_all: torch.Tensor, rows: Optional[torch.Tensor] = None,
                 want_probs: bool = False):
        """(mean cross-entropy, accuracy[, probs, prediction]) on ``cache[rows]``."""
        if self.native:
            B, probs, pred = self._fwd_native(cache, labels_all, rows, False, want_probs)
            m = self._metrics.cpu().tolist()
            out = (m[0] / B, m[1] / B)
        else:
            x = cache if rows is None else cache.index_select(0, rows.long())
            y = labels_all if rows is None else labels_all.index_select(0, rows.long())
            with torch.no_grad():
                loss, probs = self._fwd_eager(x, y)
            pred = probs.argmax(1).to(torch.int32)
            out = (float(loss), float((pred.long() == y.long()).float().mean()))
        return out + (probs, pred) if want_probs else out

    def predict(self, bottlenecks: torch.Tensor) -> torch.Tensor:
        """``final_result``: softmax probabilities for ``[B,2048]`` bottlenecks."""
        return self.activations(bottlenecks)[0]

    def activations(self, bottlenecks: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(``final_result`` probabilities, ``Wx_plus_b`` pre-activations), both [B,C], of ``[B,2048]`` bottlenecks; on a GPU
        from the head's own forward kernel (``csrc/hip/retrain_head.hip``), no library GEMM."""
        x = bottlenecks.to(self.device, torch.float32).contiguous()
        if self.native:
            B = int(x.shape[0])
            probs = torch.empty(B, self.C, dtype=torch.float32, device=self.device)
            logits = torch.empty(B, self.C, dtype=torch.float32, device=self.device)
            if B:
                labels = torch.zeros(B, dtype=torch.int32, device=self.device)
                self._fwd_native(x, labels, None, False, False, capture=(probs, logits))
            return probs, logits
        with torch.no_grad():
            logits = x @ self.weights + self.biases
            return torch.softmax(logits, 1), logits

    # ------------------------------------------------------------------ training
    def accumulate_metrics(self, cache: torch.Tensor, labels_all: torch.Tensor, rows: Optional[torch.Tensor],
                           ring_slot: int, column: int = 2, capture=None) -> None:
        """Forward only; loss sum and correct count of ``cache[rows]`` are added to ``ring[ring_slot, column:column+2]``."""
        assert self._ring is not None and column in (0, 2)
        if self.native:
            self._fwd_native(cache, labels_all, rows, False, False,
                             metrics_ptr=self._ring.data_ptr() + (ring_slot * 4 + column) * 4, capture=capture)
        else:
            x = cache if rows is None else cache.index_select(0, rows.long())
            y = labels_all if rows is None else labels_all.index_select(0, rows.long())
            with torch.no_grad():
                loss, probs = self._fwd_eager(x, y)
            if capture is not None:
                capture[0].copy_(probs)
                capture[1].copy_(x @ self.weights + self.biases)
            self._ring[ring_slot, column] += loss * x.shape[0]
            self._ring[ring_slot, column + 1] += (probs.argmax(1) == y.long()).sum()

    def train_step(self, cache: torch.Tensor, labels_all: torch.Tensor, rows: Optional[torch.Tensor] = None,
                   group=None, ring_slot: Optional[int] = None, capture=None) -> Tuple[float, float]:
        """One SGD step on ``cache[rows]``; returns (cross-entropy, accuracy) of the batch *before* the update
        as device-synchronising floats only when asked through ``last_metrics``."""
        use_ring = ring_slot is not None and self._ring is not None
        if self.native:
            B, _, _ = self._fwd_native(cache, labels_all, rows, True, False,
                                       metrics_ptr=self._ring.data_ptr() + ring_slot * 16 if use_ring else 0, capture=capture)
            st = torch.cuda.current_stream().cuda_stream
            lib = native.hip()
            if self.world == 1 and self.comm is None:
                native.check(lib.dtf_retrain_head_bwd(cache.data_ptr(), native.ptr(rows), self._dz.data_ptr(), B, self.C,
                                                      self.P.data_ptr(), self.P.data_ptr() + 4 * BOTTLENECK * self.C, 0, 0,
                                                      self.lr, st), "head bwd")
            else:
                from ..ops import hip_ops
                from ..parallel.ddp import all_reduce_
                gW = self.G[:BOTTLENECK * self.C]
                gb = self.G[BOTTLENECK * self.C:]
                native.check(lib.dtf_retrain_head_bwd(cache.data_ptr(), native.ptr(rows), self._dz.data_ptr(), B, self.C,
                                                      0, 0, gW.data_ptr(), gb.data_ptr(), 0.0, st), "head bwd")
                if self.comm is not None:
                    self.comm.all_reduce(self.G)  # (ncclAllReduce on this stream: stream-ordered between bwd and the update)
                else:
                    all_reduce_(self.G, group=group)
                hip_ops.sgd_(self.P, self.G, self._step_t, self._ticket, lr=self.lr, grad_scale=1.0 / self.world)
        else:
            x = cache if rows is None else cache.index_select(0, rows.long())
            y = labels_all if rows is None else labels_all.index_select(0, rows.long())
            leaf = self.P.detach().clone().requires_grad_(True)
            w = leaf[:BOTTLENECK * self.C].view(BOTTLENECK, self.C)
            b = leaf[BOTTLENECK * self.C:BOTTLENECK * self.C + self.C]
            logits = x @ w + b
            p = torch.softmax(logits, 1)
            if capture is not None:
                capture[0].copy_(p.detach())
                capture[1].copy_(logits.detach())
            z = p if self.double_softmax else logits
            loss = torch.nn.functional.nll_loss(torch.log_softmax(z, 1), y.long())
            (g,) = torch.autograd.grad(loss, leaf)
            if self.world > 1:
                from ..parallel.ddp import all_reduce_
                all_reduce_(g, group=group)
                g = g / self.world
            self.P.sub_(g, alpha=self.lr)
            self._metrics[0] = loss.detach() * x.shape[0]
            self._metrics[1] = (p.argmax(1) == y.long()).sum()
            if use_ring:
                self._ring[ring_slot, 0] += self._metrics[0]
                self._ring[ring_slot, 1] += self._metrics[1]
            B = x.shape[0]
        self.global_step += 1
        self._last_B = B
        return B

    def last_metrics(self) -> Tuple[float, float]:
        m = self._metrics.cpu().tolist()
        return m[0] / self._last_B, m[1] / self._last_B


class TinyBackbone:
    """A deterministic stand-in feature extractor (8x8 average pooling + fixed random projection + ReLU
    -> 2048-d) for CPU plumbing tests where a 11.4 GFLOP/image Inception forward is not affordable."""

    def __init__(self, device="cpu", seed: int = 0):
        self.device = torch.device(device)
        g = torch.Generator().manual_seed(seed)
        self.proj = (torch.randn(8 * 8 * 3, BOTTLENECK, generator=g) / 8.0).to(self.device)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        # x: [B,299,299,8] bf16/fp32 NHWC (channels 0..2 real)
        v = x[..., :3].float().permute(0, 3, 1, 2)
        pooled = torch.nn.functional.adaptive_avg_pool2d(v, 8).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        return torch.relu(pooled @ self.proj)
