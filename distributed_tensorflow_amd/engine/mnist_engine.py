"""MNIST CNN training engines.

``NativeMnistEngine`` — the MI355X hot path.  It owns a flat fp32 parameter arena
(``P``/``G``/``M``/``V``: weights, gradients, Adam slots; TF layouts, arena order =
gradient-readiness order), the bf16 activation workspaces and a private HIP stream, and
drives the C++ step executor (``csrc/hip/engine.hip``): 8 kernels per train step, replayed as a
hipGraph (16 steps per graph launch).  ``dtype="fp32"`` runs the whole step at the reference's
precision (fp32 activations, exact-fp32 MFMA); ``dtype="bf16"`` uses bf16 MFMA operands.  Batches are gathered on-device from an HBM-resident
dataset through a device permutation; loss/accuracy land in a device ring buffer that is
read back only at logging points (the reference syncs every step, quirk Q5).

``EagerMnistEngine`` — the same interface on plain PyTorch ops (``models.mnist_cnn``):
CPU plumbing config (BASELINE config 1), the correctness oracle, and the measured
"PyTorch-ROCm eager" baseline.

Reference: ``demo1/train.py:49-165`` (model, loss, Adam 1e-4, keep_prob 0.7, eval).
"""
from __future__ import annotations

import os

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import native
from ..data.mnist import DeviceDataset
from ..models import mnist_cnn
from ..optim.adam_tf import adam_tf_flat_

ARENA_ALIGN = 64  # elements (256 B): keeps every segment float4 / cache-line aligned


def arena_layout() -> Tuple["OrderedDict[str, Tuple[int, int, tuple]]", int]:
    """name -> (offset, numel, shape); total padded element count."""
    off = 0
    out: "OrderedDict[str, Tuple[int, int, tuple]]" = OrderedDict()
    for name, (_tf, shape, _k) in mnist_cnn.PARAM_SPECS.items():
        n = mnist_cnn.numel(shape)
        out[name] = (off, n, shape)
        off += (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    return out, off


class _EngineBase:
    double_softmax: bool
    batch_size: int

    def param_views(self) -> "OrderedDict[str, torch.Tensor]":
        raise NotImplementedError

    def export_numpy(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().float().cpu().numpy().copy() for k, v in self.param_views().items()}


class NativeMnistEngine(_EngineBase):
    RING = 1024

    def __init__(self, dataset: Optional[DeviceDataset], *, batch_size: int = 100, lr: float = 1e-4,
                 keep_prob: float = 0.7, double_softmax: bool = True, seed: Optional[int] = None,
                 world: int = 1, eval_chunk: int = 5000, use_graph: bool = True,
                 fc1_splits: int = 14, bwd_splits: int = 2, dw2_splits: int = 5, head_blocks: int = 0,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, device="cuda", dtype: str = "bf16"):
        if not torch.cuda.is_available():
            raise native.NativeError("NativeMnistEngine needs a GPU (MI355X); use EagerMnistEngine on CPU")
        bwd_splits = int(os.environ.get("DTF_BWD_SPLITS", bwd_splits))  # (tuning knobs for A/B runs of bench.py)
        fc1_splits = int(os.environ.get("DTF_FC1_SPLITS", fc1_splits))
        dw2_splits = int(os.environ.get("DTF_DW2_IPW", dw2_splits))
        head_blocks = int(os.environ.get("DTF_HEAD_BLOCKS", head_blocks))
        if dtype not in ("bf16", "fp32"):
            raise ValueError("dtype must be bf16 or fp32")
        # fp32 = the reference's precision (demo1/train.py computes in float32): fp32 activations and activation gradients,
        # every GEMM-shaped op on the exact-fp32 MFMA.  bf16 = bf16 MFMA operands, fp32 accumulation and master weights.
        self.dtype = dtype
        f32 = dtype == "fp32"
        if f32:  # the fp32 kernels' own tile geometry (f32_fc.hip): 14 K-slabs of 224 forward, 6 of <= 192 backward
            # (backward 4 -> 6: the data-gradient workgroups of the fused fc backward launch are its longest; 143.8 -> 142.0 us
            # per step, scripts/gpu_sweep_knobs.sh; 8 and more lose again)
            fc1_splits, bwd_splits = int(os.environ.get("DTF_FC1_SPLITS", 14)), int(os.environ.get("DTF_BWD_SPLITS", 6))
        self.lib = native.hip()
        native.check(self.lib.dtf_conv2_lds_init() or self.lib.dtf_conv2_bwd_lds_init() or self.lib.dtf_conv2_f32_init(),
                     "LDS kernel attribute setup")  # before any stream capture
        self.device = torch.device(device)
        self.dataset = dataset
        self.batch_size = batch_size
        self.double_softmax = double_softmax
        self.world = world
        self.use_graph = use_graph
        self.maxB = max(batch_size, eval_chunk)
        self.stream = torch.cuda.Stream(device=self.device)
        self.layout, self.n_arena = arena_layout()
        dev = self.device
        is_f32, f32 = f32, dict(dtype=torch.float32, device=dev)
        self.P = torch.zeros(self.n_arena, **f32)
        self.G = torch.zeros(self.n_arena, **f32)
        self.M = torch.zeros(self.n_arena, **f32)
        self.V = torch.zeros(self.n_arena, **f32)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        # [skip flag, skipped-step count]: raised by the IPC gradient exchange when one of its waits expired, honoured and
        # cleared by the optimizer launch (a step with an incomplete gradient sum is never applied)
        self.skip_t = torch.zeros(2, dtype=torch.int32, device=dev)
        self.metrics_ring = torch.zeros(self.RING * 2, **f32)
        self.eval_metrics = torch.zeros(2, **f32)
        mb = self.maxB
        bf = f32 if is_f32 else dict(dtype=torch.bfloat16, device=dev)  # activations / activation gradients
        u8 = dict(dtype=torch.uint8, device=dev)
        self.ws = {
            "pool1": torch.zeros(mb * 196 * 32, **bf), "amax1": torch.zeros(mb * 196 * 32, **u8),
            "pool2": torch.zeros(mb * 49 * 64, **bf), "amax2": torch.zeros(mb * 49 * 64, **u8),
            "slabs": torch.zeros(max(fc1_splits * batch_size, mb) * 1024, **f32),
            "h_drop": torch.zeros(batch_size * 1024, **bf), "dh": torch.zeros(batch_size * 1024, **bf),
            "dz": torch.zeros(batch_size * 16, **bf),
            "dpool2": torch.zeros(batch_size * 3136, **f32), "dy2": torch.zeros(batch_size * 196 * 64, **bf),
            "dpool1": torch.zeros(batch_size * 196 * 32, **f32),
            # packed bf16 copies of the conv2 filter: the two LDS layouts of the bf16 path (csrc/hip/conv2_pack.h), or the
            # three bf16 planes per operand of the fp32 path's bf16x6 conv kernels (csrc/hip/conv2_pack3.h)
            "wc2f": torch.zeros((3 if is_f32 else 1) * 800 * 64, dtype=torch.bfloat16, device=dev),
            "wc2b": torch.zeros((3 if is_f32 else 1) * 800 * 64, dtype=torch.bfloat16, device=dev),
        }
        self.seed = int(seed if seed is not None else np.random.SeedSequence().entropy % (2 ** 62))
        self.spe = dataset.steps_per_epoch(batch_size) if dataset is not None else 1
        self._steps_done = 0  # host mirror of the device step counter
        s = native.MnistEngineStruct()
        s.B, s.maxB, s.spe, s.world = batch_size, self.maxB, self.spe, world
        s.double_softmax = int(double_softmax)
        s.fc1_splits, s.bwd_splits, s.dw2_splits = fc1_splits, bwd_splits, dw2_splits
        # one workgroup per row: the rows' latency chains run side by side (50 -> 100 workgroups: 108.2 -> 105.6 us per step)
        s.head_blocks = head_blocks if head_blocks > 0 else min(batch_size, 1024)
        s.ring = self.RING
        s.lr, s.beta1, s.beta2, s.eps, s.keep_prob = lr, beta1, beta2, eps, keep_prob
        s.seed = self.seed
        s.f32 = int(is_f32)
        s.skip = self.skip_t.data_ptr()
        if dataset is not None:
            s.images, s.labels, s.perm = dataset.images.data_ptr(), dataset.labels.data_ptr(), dataset.perm.data_ptr()
        s.P, s.G, s.M, s.V = (t.data_ptr() for t in (self.P, self.G, self.M, self.V))
        s.n_arena = self.n_arena
        for field, name in (("off_Wfc1", "W_fc1"), ("off_bfc1", "b_fc1"), ("off_Wfc2", "W_fc2"), ("off_bfc2", "b_fc2"),
                            ("off_Wc2", "W_conv2"), ("off_bc2", "b_conv2"), ("off_Wc1", "W_conv1"), ("off_bc1", "b_conv1")):
            setattr(s, field, self.layout[name][0])
        s.step, s.ticket, s.metrics_ring = self.step_t.data_ptr(), self.ticket.data_ptr(), self.metrics_ring.data_ptr()
        for k in ("pool1", "amax1", "pool2", "amax2", "slabs", "h_drop", "dh", "dz", "dpool2", "dy2", "dpool1", "wc2f", "wc2b"):
            setattr(s, k, self.ws[k].data_ptr())
        self.cfg = s
        self._graphs = {}
        self.grad_hook = None
        self.state_provider = None   # set by the data-parallel trainer while the live arenas are not a consistent state
        self.on_state_loaded = None  # called after the arenas were rewritten from outside (restore, re-initialisation)
        # stream capture vs. snapshots from other threads (the Supervisor's autosave thread calls state_dict()): work
        # enqueued on a capturing stream would be recorded into the step graph, and synchronising it invalidates the capture
        import threading
        self._capture_lock = threading.RLock()
        torch.cuda.synchronize(self.device)  # allocations/zero-fills ran on the default stream
        self.init_params(seed)

    # ------------------------------------------------------------------ parameters / state
    def param_views(self, arena: Optional[torch.Tensor] = None) -> "OrderedDict[str, torch.Tensor]":
        a = self.P if arena is None else arena
        return OrderedDict((k, a[o:o + n].view(shape)) for k, (o, n, shape) in self.layout.items())

    def init_params(self, seed: Optional[int] = None) -> None:
        self.load_params(mnist_cnn.init_params(seed))

    def load_params(self, params: Dict[str, torch.Tensor]) -> None:
        with torch.cuda.stream(self.stream):
            for k, v in self.param_views().items():
                v.copy_(torch.as_tensor(params[k]).to(self.device, torch.float32).reshape(v.shape))
        self.stream.synchronize()
        if getattr(self, "on_state_loaded", None) is not None:
            self.on_state_loaded()

    def bucket_bounds(self) -> Tuple[Tuple[int, int], Tuple[int, int]]:
        """Gradient buckets in readiness order: A = fc layers (after part 1), B = conv layers (after part 2)."""
        split = self.layout["W_conv2"][0]
        return (0, split), (split, self.n_arena)

    def fc1_grad_numel(self) -> int:
        """Size of the fc1 filter gradient, the stored (never atomically accumulated) segment at the arena's head."""
        off, n, _shape = self.layout["W_fc1"]
        assert off == 0
        return int(n)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        # snapshot in stream order (between two steps) so a saver thread never sees a half-applied update; never while
        # the training thread has a capture open on the stream
        if self.state_provider is not None:  # (optimizer state sharded over the ranks: the trainer's last consistent snapshot)
            return self.state_provider()
        with self._capture_lock:
            with torch.cuda.stream(self.stream):
                sd = {"P": self.P.clone(), "M": self.M.clone(), "V": self.V.clone(), "step": self.step_t.clone()}
            self.stream.synchronize()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.cuda.stream(self.stream):
            self.P.copy_(sd["P"].to(self.device))
            self.M.copy_(sd["M"].to(self.device))
            self.V.copy_(sd["V"].to(self.device))
            self.step_t.copy_(sd["step"].to(self.device).to(torch.int32))
        self.stream.synchronize()
        self._steps_done = int(self.step_t.item())
        if self.on_state_loaded is not None:
            self.on_state_loaded()

    @property
    def global_step(self) -> int:
        return self._steps_done

    def reset_training(self, dataset: DeviceDataset, seed: Optional[int] = None) -> None:
        """Start a fresh run on another device-resident split with this engine's buffers, kernels and gradient hooks:
        re-initialised parameters (the same on every rank for the same ``seed``), zeroed optimizer state, step counter and
        metric ring.  Drops the captured graphs (they hold the old split's addresses)."""
        self.stream.synchronize()
        self.dataset = dataset
        self.spe = dataset.steps_per_epoch(self.batch_size)
        self.cfg.spe = self.spe
        self.cfg.images, self.cfg.labels, self.cfg.perm = (dataset.images.data_ptr(), dataset.labels.data_ptr(),
                                                           dataset.perm.data_ptr())
        with torch.cuda.stream(self.stream):
            for t in (self.G, self.M, self.V, self.step_t, self.ticket, self.metrics_ring):
                t.zero_()
        self.stream.synchronize()
        self._steps_done = 0
        self.init_params(seed)
        self._drop_graph()

    def skipped_steps(self) -> int:
        """Optimizer launches that did not apply their step because a wait of the gradient exchange had expired
        (synchronises).  The device step counter does not advance on such a step; ``resync_step_counter`` realigns the
        host mirror."""
        self.stream.synchronize()
        return int(self.skip_t[1].item())

    def resync_step_counter(self) -> int:
        self.stream.synchronize()
        self._steps_done = int(self.step_t.item())
        return self._steps_done

    def set_lr(self, lr: float) -> None:
        self.cfg.lr = lr
        self._drop_graph()

    def set_world(self, world: int) -> None:
        self.world = world
        self.cfg.world = world
        self._drop_graph()

    def _drop_graph(self) -> None:
        for h in self._graphs.values():
            self.lib.dtf_graph_free(h)
        self._graphs = {}

    # ------------------------------------------------------------------ stepping
    def _maybe_reshuffle(self) -> None:
        if self.dataset is not None and self._steps_done > 0 and self._steps_done % self.spe == 0:
            with torch.cuda.stream(self.stream):
                self.dataset.reshuffle()

    def fwd_bwd(self, parts: int = 3) -> None:
        native.check(self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), parts, self.stream.cuda_stream), "mnist fwd/bwd")

    def apply(self) -> None:
        native.check(self.lib.dtf_mnist_apply(C.byref(self.cfg), self.stream.cuda_stream), "mnist apply")
        self._steps_done += 1

    def _capture(self, key: str) -> None:
        """Capture one segment of the step into a hipGraph: ``full`` (fwd+bwd+Adam), ``part1``
        (forward + head + fc grads), ``part2`` (conv backward) or ``apply`` (Adam)."""
        st = self.stream.cuda_stream
        with self._capture_lock:
            self.stream.synchronize()
            native.check(self.lib.dtf_graph_begin(st), "graph begin")
            rc = self._enqueue(key, st)
            h = C.c_void_p()
            rc2 = self.lib.dtf_graph_end(st, C.byref(h))
            native.check(rc or rc2, "graph capture (%s)" % key)
            self._graphs[key] = h

    def _enqueue(self, key: str, st: int) -> int:
        """Enqueue the kernels of one step segment on stream ``st`` (directly or under capture)."""
        if key == "full":
            hook = self.grad_hook
            # W_fc1's Adam update inside the conv filter-gradient launch, wherever this rank holds the global fc1 gradient
            early = 4 if self.early_fc1_adam() else 0
            if hook is None:
                rc = self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), 3 | early, st)
            else:
                # in-stream gradient collective (the IPC all-reduce is plain kernels, so it is captured into the same
                # hipGraph and the whole data-parallel step stays ONE graph launch): the fc gradients are reduced on a
                # forked stream while the conv backward runs, the conv gradients after it
                rc = self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), 1, st)
                if rc == 0:
                    hook.after_part1(self.stream)
                    rc = self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), 2 | early, st)
                if rc == 0:
                    hook.after_part2(self.stream)
            comm = getattr(hook, "apply_comm", None)
            if rc == 0 and comm is not None:  # gather schedule: the optimizer launch reduces the published gradients itself
                if not early:
                    return -22  # (only the arena after W_fc1 is published; its own update must have happened above)
                return self.lib.dtf_mnist_apply_part_ipc(C.byref(self.cfg), 2, C.byref(comm), st)
            return rc or self.lib.dtf_mnist_apply_part(C.byref(self.cfg), 2 if early else 0, st)
        if key.startswith("full*"):  # several consecutive steps in one graph (see train_steps)
            for _ in range(int(key[5:])):
                rc = self._enqueue("full", st)
                if rc:
                    return rc
            return 0
        if key == "part1":
            return self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), 1, st)
        if key == "part2":
            return self.lib.dtf_mnist_fwd_bwd(C.byref(self.cfg), 2, st)
        if key == "apply":
            return self.lib.dtf_mnist_apply(C.byref(self.cfg), st)
        return -22

    def set_grad_redirect(self, fc1_dst: int = 0, apply_src: int = 0) -> None:
        """Zero-copy gradient exchange: store the fc1 filter gradient (arena offset 0, 98 % of the bytes) at device
        address ``fc1_dst`` instead of into ``G``, and let the optimizer read the whole arena's gradients from
        ``apply_src`` (0 = back to ``G`` for both).  Drops the captured graphs."""
        self.cfg.G_fc1_dst = fc1_dst or None
        self.cfg.G_src = apply_src or None
        self._drop_graph()

    def early_fc1_adam(self) -> bool:
        """Whether the single-launch step updates W_fc1 (98 % of the optimizer's bytes) inside the conv filter-gradient
        launch, next to those latency-bound kernels, instead of in the optimizer launch after them: one replica (no
        gradient hook), or the gather schedule, where every rank computes the global fc1 gradient itself.
        ``DTF_EARLY_FC1_ADAM=0`` switches it off."""
        if os.environ.get("DTF_EARLY_FC1_ADAM", "1") == "0":
            return False
        if self.cfg.gather_n > 0:
            return True
        return self.grad_hook is None and self.world == 1 and not self.cfg.G_fc1_dst and not self.cfg.G_src

    def gather_supported(self) -> bool:
        """The activation-gather schedule rides on the fused fc backward launch (split-K data gradient; bf16: batch <= 128,
        fp32: the batch a multiple of the filter-gradient GEMM's K slab of 20, so a slab never straddles two ranks)."""
        if self.cfg.bwd_splits <= 1:
            return False
        return self.batch_size % 20 == 0 if self.dtype == "fp32" else self.batch_size <= 128

    def gather_bytes(self) -> Tuple[int, int]:
        """Bytes of the two fc1 filter-gradient operands a rank publishes: pool2 [B,3136] and dh [B,1024] in the compute
        dtype."""
        es = 4 if self.dtype == "fp32" else 2
        return self.batch_size * 3136 * es, self.batch_size * 1024 * es

    def set_gather(self, ipc=None, early_push: bool = False) -> None:
        """Data-parallel "gather" schedule: a training step keeps its pool2 / dh in this rank's stage buffer of ``ipc``
        (an IpcAllReduce whose stage buffers are at least gather_bytes() large), ranks meet after the loss head, and the
        fc1 filter gradient is computed over ALL ranks' operands, read in place from the peers' buffers; ``None``
        switches it off.  Drops the captured graphs."""
        self.cfg.early_push = 0
        if ipc is None:
            self.cfg.gather_n = 0
            self.cfg.gather_comm = None
            self.cfg.gather_local = None
            self._gather_buf = None
        else:
            b0, b1 = self.gather_bytes()
            assert (b0 + b1) // 4 <= ipc.n_max and self.gather_supported()
            # the peers' operands are pulled into this buffer once per step (peer memory is not cached by the reader's L2,
            # and the GEMM re-reads every tile from many workgroups); this rank's own are read from its stage buffer
            self._gather_buf = torch.empty(ipc.world * (b0 + b1), dtype=torch.uint8, device=self.device)
            self.cfg.gather_local = self._gather_buf.data_ptr()
            self.cfg.gather_bytes = b0 + b1
            for p in range(ipc.world):
                base = ipc.peer_stage_ptr(p) if p == ipc.rank else self._gather_buf.data_ptr() + p * (b0 + b1)
                self.cfg.gather_A[p] = base
                self.cfg.gather_B[p] = base + b0
            self.cfg.gather_comm = ipc.comm_addr
            self.cfg.gather_n = ipc.world
            self.cfg.gather_rank = ipc.rank
            if early_push:
                # "ipc-gather2": every rank stores its pool2 into block `rank` of every peer's result buffer during the fc1
                # forward launch; the exchange after the loss head then only moves dh (csrc/hip/engine.hip: early_push)
                assert self.dtype == "fp32" and ipc.n_max >= ipc.world * (b0 // 4), "exchange buffer too small for the early push"
                self.cfg.early_push = 1
        self._drop_graph()

    @staticmethod
    def shard_rows(rank: int, world: int, rows: int = 3136) -> Tuple[int, int]:
        """Rows [lo, hi) of the fc1 filter gradient that ``rank`` of ``world`` computes under the shard schedule (multiples of
        4: the GEMM's operand alignment; 3136 = 4 x 784 splits evenly for 2 / 4 / 7 / 8 ranks)."""
        q = rows // 4
        return (q * rank // world) * 4, (q * (rank + 1) // world) * 4

    def shard_supported(self) -> bool:
        """The shard schedule is built from the exact-fp32 kernels and rides on the gather schedule's operand exchange."""
        return self.dtype == "fp32" and self.gather_supported()

    def set_shard(self, ipc_act=None, ipc_w=None) -> None:
        """Data-parallel "shard" schedule (``csrc/hip/shard_body.h``): per-rank work independent of the world size.  The
        operands travel as under ``set_gather`` (``ipc_act``), but each rank pulls only the pool2 columns of ITS rows of
        the fc1 filter gradient, computes those rows of the global gradient, and pushes them into every rank's landing
        buffer (the result buffers of ``ipc_w``, an IpcAllReduce of at least fc1_grad_numel() floats), from where W_fc1's
        Adam update reads.  ``None`` switches it off (and the gather exchange with it).  Drops the captured graphs.
        The caller zeroes G's fc1 segment before the first shard step (the slice is accumulated with atomics)."""
        if ipc_act is None or ipc_w is None:
            self.cfg.shard_n = 0
            self.cfg.wshard = 0
            self.cfg.wshard_rows_a = 0
            self.cfg.shard_comm = None
            self.set_gather(None)
            return
        assert self.shard_supported() and ipc_w.n_max >= self.fc1_grad_numel() and ipc_w.world == ipc_act.world
        self.set_gather(ipc_act)
        lo, hi = self.shard_rows(ipc_act.rank, ipc_act.world)
        self.cfg.shard_comm = ipc_w.comm_addr
        self.cfg.shard_n, self.cfg.shard_lo, self.cfg.shard_w = ipc_act.world, lo, hi - lo
        self.cfg.wshard = 0
        self.cfg.wshard_rows_a = 0
        self._drop_graph()

    def set_wshard(self, ipc_act, ipc_w, split: float = 0.0) -> None:
        """Data-parallel "wshard" schedule (``csrc/hip/engine.hip``, ``MnistEngine::wshard``): as ``set_shard``, but the
        slices that travel are the updated WEIGHTS.  W_fc1 lives in the two buffers of ``ipc_w`` (alternating by step
        parity) instead of in ``P``; this rank applies Adam to its rows only and stores them into every rank's buffer of the
        next parity.  ``P[W_fc1]`` and the optimizer state outside this rank's rows go stale until ``wshard_store()`` / the
        trainer's ``materialize()``.  The caller loads the buffers (``wshard_load``) before the first step.
        ``split``: fraction of this rank's rows whose update (and stores to the peers) rides in the conv2 data-gradient
        launch instead of the filter-gradient launch, so that the stores have two launches to drain (2 and 4 GPUs)."""
        self.set_shard(ipc_act, ipc_w)
        self.cfg.wshard = 1
        self.cfg.wshard_rows_a = min(self.cfg.shard_w - 4, max(0, int(self.cfg.shard_w * split) // 4 * 4))

    def wshard_load(self) -> None:
        """P[W_fc1] -> both parity buffers of this rank (every replica holds the same P when this is called)."""
        native.check(self.lib.dtf_mnist_wshard_load(C.byref(self.cfg), self.stream.cuda_stream), "wshard load")

    def wshard_store(self) -> None:
        """The buffer of the current step parity -> P[W_fc1] (stream-ordered, local)."""
        native.check(self.lib.dtf_mnist_wshard_store(C.byref(self.cfg), self.stream.cuda_stream), "wshard store")

    def set_grad_hook(self, hook) -> None:
        """``hook.after_part1(stream)`` / ``hook.after_part2(stream)`` run inside the ``full`` segment after the fc
        gradients resp. all gradients have been enqueued (``stream`` is the engine's torch stream)."""
        self.grad_hook = hook
        self._drop_graph()

    def launch(self, key: str) -> None:
        """Run a step segment (graph replay when ``use_graph``).  ``apply``/``full`` advance the step."""
        if key in ("full", "part1"):
            self._maybe_reshuffle()
        if self.use_graph:
            if key not in self._graphs:
                self._capture(key)
            native.check(self.lib.dtf_graph_launch(self._graphs[key], self.stream.cuda_stream), "graph launch")
        else:
            native.check(self._enqueue(key, self.stream.cuda_stream), "mnist " + key)
        if key in ("full", "apply"):
            self._steps_done += 1

    def train_step(self) -> None:
        """One full step (single-GPU: fwd+bwd+Adam as one hipGraph replay)."""
        self.launch("full")

    def train_steps(self, n: int, chunk: int = 16) -> None:
        """``n`` full steps.  Everything a step needs (batch offset, dropout counter, Adam step, metric slot) lives in
        device memory, so ``chunk`` consecutive steps replay as ONE hipGraph launch: the ~6 us the command processor
        needs between two graph launches is paid once per chunk (8 -> 16 steps per launch: 91.2 -> 90.7 us per step).  A chunk never spans an
        epoch boundary, where the permutation is regenerated between steps."""
        chunk = int(os.environ.get("DTF_STEPS_PER_GRAPH", chunk))
        if self.dataset is not None:
            chunk = min(chunk, self.spe)  # (a data set with fewer steps per epoch than a chunk still gets one graph per epoch)
        key = "full*%d" % chunk
        while n > 0:
            room = self.spe - (self._steps_done % self.spe) if self.dataset is not None else n
            if self.use_graph and chunk > 1 and n >= chunk and room >= chunk:
                self._maybe_reshuffle()
                if key not in self._graphs:
                    self._capture(key)
                native.check(self.lib.dtf_graph_launch(self._graphs[key], self.stream.cuda_stream), "graph launch")
                self._steps_done += chunk
                n -= chunk
            else:
                self.train_step()
                n -= 1

    def prepare(self, chunk: int = 16) -> None:
        """Capture the step graphs now (capturing records, it does not execute) so that no later step pays for it."""
        if self.use_graph:
            chunk = int(os.environ.get("DTF_STEPS_PER_GRAPH", chunk))
            if self.dataset is not None:
                chunk = min(chunk, self.spe)
            for key in ("full",) + (("full*%d" % chunk,) if chunk > 1 else ()):
                if key not in self._graphs:
                    self._capture(key)

    def graph_nodes(self) -> int:
        return sum(self.lib.dtf_graph_num_nodes(h) for h in self._graphs.values()) if self._graphs else -1

    # ------------------------------------------------------------------ metrics / eval
    def read_metrics(self, first_step: int, last_step: int) -> Tuple[np.ndarray, np.ndarray]:
        """(loss mean, accuracy) per step for device steps [first_step, last_step) (<= RING steps back)."""
        self.stream.synchronize()
        ring = self.metrics_ring.view(self.RING, 2).cpu().numpy()
        idx = np.arange(first_step, last_step) % self.RING
        return ring[idx, 0] / self.batch_size, ring[idx, 1] / self.batch_size

    def evaluate(self, images: torch.Tensor, labels: torch.Tensor) -> Tuple[float, float]:
        """Mean loss and accuracy over a device-resident split (fp32 [N,784], int32 [N]); chunked on device."""
        n = images.shape[0]
        assert images.dtype == torch.float32 and labels.dtype == torch.int32 and images.is_contiguous()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.eval_metrics.zero_()
            native.check(self.lib.dtf_mnist_eval(C.byref(self.cfg), images.data_ptr(), labels.data_ptr(), n,
                                                 self.eval_metrics.data_ptr(), 0, 0, self.stream.cuda_stream), "mnist eval")
        self.stream.synchronize()
        m = self.eval_metrics.cpu().numpy()
        return float(m[0]) / n, float(m[1]) / n

    def predict(self, images: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """softmax probabilities [N,10] and argmax [N] for fp32 [N,784] device images."""
        images = images.to(self.device, torch.float32).contiguous()
        n = images.shape[0]
        probs = torch.empty(n, 10, dtype=torch.float32, device=self.device)
        pred = torch.empty(n, dtype=torch.int32, device=self.device)
        labels = torch.zeros(n, dtype=torch.int32, device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.eval_metrics.zero_()
            native.check(self.lib.dtf_mnist_eval(C.byref(self.cfg), images.data_ptr(), labels.data_ptr(), n,
                                                 self.eval_metrics.data_ptr(), probs.data_ptr(), pred.data_ptr(),
                                                 self.stream.cuda_stream), "mnist predict")
        self.stream.synchronize()
        return probs, pred

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:
            pass


class EagerMnistEngine(_EngineBase):
    """Same interface on PyTorch ops (CPU or GPU)."""

    RING = 1024

    def __init__(self, dataset, *, batch_size: int = 100, lr: float = 1e-4, keep_prob: float = 0.7,
                 double_softmax: bool = True, seed: Optional[int] = None, world: int = 1, device="cpu",
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, philox_seed: Optional[int] = None,
                 **_unused):
        self.device = torch.device(device)
        self.dataset = dataset
        self.batch_size, self.lr, self.keep_prob = batch_size, lr, keep_prob
        self.double_softmax, self.world = double_softmax, world
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.layout, self.n_arena = arena_layout()
        self.P = torch.zeros(self.n_arena, device=self.device)
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self._steps_done = 0
        self.spe = dataset.steps_per_epoch(batch_size) if dataset is not None else 1
        self._ring = np.zeros((self.RING, 2), np.float64)
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(seed if seed is not None else int(np.random.SeedSequence().entropy % (2 ** 31)))
        # philox_seed = a NativeMnistEngine's ``seed``: draw exactly that engine's dropout masks (utils/philox.py)
        self.philox_seed = philox_seed
        self.load_params(mnist_cnn.init_params(seed))

    def param_views(self, arena=None):
        a = self.P if arena is None else arena
        return OrderedDict((k, a[o:o + n].view(shape)) for k, (o, n, shape) in self.layout.items())

    def load_params(self, params):
        for k, v in self.param_views().items():
            v.copy_(torch.as_tensor(params[k]).to(self.device, torch.float32).reshape(v.shape))

    def init_params(self, seed=None):
        self.load_params(mnist_cnn.init_params(seed))

    def bucket_bounds(self):
        split = self.layout["W_conv2"][0]
        return (0, split), (split, self.n_arena)

    def state_dict(self):
        return {"P": self.P.clone(), "M": self.M.clone(), "V": self.V.clone(),
                "step": torch.tensor([self._steps_done], dtype=torch.int32)}

    def load_state_dict(self, sd):
        self.P.copy_(sd["P"].to(self.device))
        self.M.copy_(sd["M"].to(self.device))
        self.V.copy_(sd["V"].to(self.device))
        self._steps_done = int(sd["step"].reshape(-1)[0])

    @property
    def global_step(self) -> int:
        return self._steps_done

    def set_lr(self, lr):
        self.lr = lr

    def set_world(self, world):
        self.world = world

    def _batch(self):
        if self._steps_done > 0 and self._steps_done % self.spe == 0:
            self.dataset.reshuffle()
        idx = self.dataset.batch_indices(self._steps_done % self.spe, self.batch_size)
        return self.dataset.gather(idx)

    def fwd_bwd(self, parts: int = 3, batch=None) -> None:
        if not parts & 1:
            return  # the eager backward is monolithic: everything happens in part 1
        x, y = self._batch() if batch is None else batch
        leaf = self.P.detach().clone().requires_grad_(True)
        views = self.param_views(leaf)
        mask = None
        if self.keep_prob < 1.0 and self.philox_seed is not None:
            from ..utils.philox import dropout_keep_mask
            mask = torch.from_numpy(dropout_keep_mask(x.shape[0], self._steps_done, self.philox_seed, self.keep_prob)).to(self.device)
        elif self.keep_prob < 1.0:
            mask = (torch.rand(x.shape[0], 1024, generator=self._gen) < self.keep_prob).float().to(self.device)
        logits = mnist_cnn.forward_logits(views, x, keep_prob=self.keep_prob, dropout_mask=mask, training=True)
        loss, pred = mnist_cnn.softmax_xent(logits, y.long(), double_softmax=self.double_softmax)
        (g,) = torch.autograd.grad(loss, leaf)
        self.G.copy_(g)
        slot = self._steps_done % self.RING
        self._ring[slot, 0] = float(loss.detach()) * x.shape[0]
        self._ring[slot, 1] = float((pred.argmax(1) == y.long()).sum())

    def apply(self) -> None:
        adam_tf_flat_(self.P, self.G, self.M, self.V, self._steps_done + 1, self.lr, self.b1, self.b2, self.eps,
                      grad_scale=1.0 / max(1, self.world))
        self.G.zero_()
        self._steps_done += 1

    def train_step(self) -> None:
        self.fwd_bwd(3)
        self.apply()

    def read_metrics(self, first_step, last_step):
        idx = np.arange(first_step, last_step) % self.RING
        return self._ring[idx, 0] / self.batch_size, self._ring[idx, 1] / self.batch_size

    @torch.no_grad()
    def evaluate(self, images, labels, chunk: int = 5000):
        n = images.shape[0]
        loss_sum, correct = 0.0, 0.0
        views = self.param_views()
        for s in range(0, n, chunk):
            x, y = images[s:s + chunk].to(self.device), labels[s:s + chunk].to(self.device).long()
            logits = mnist_cnn.forward_logits(views, x)
            loss, pred = mnist_cnn.softmax_xent(logits, y, double_softmax=self.double_softmax)
            loss_sum += float(loss) * x.shape[0]
            correct += float((pred.argmax(1) == y).sum())
        return loss_sum / n, correct / n

    @torch.no_grad()
    def predict(self, images):
        logits = mnist_cnn.forward_logits(self.param_views(), images.to(self.device, torch.float32))
        p = torch.softmax(logits, 1)
        return p, p.argmax(1).to(torch.int32)


def make_engine(kind: str, dataset, **kw):
    """``auto`` -> native on a GPU, eager otherwise."""
    if kind == "auto":
        kind = "native" if torch.cuda.is_available() else "eager"
    if kind == "native":
        return NativeMnistEngine(dataset, **kw)
    if kind == "eager":
        kw.pop("use_graph", None)
        kw.pop("eval_chunk", None)
        kw.pop("dtype", None)  # (plain PyTorch fp32)
        return EagerMnistEngine(dataset, **kw)
    raise ValueError("unknown engine %r" % kind)
