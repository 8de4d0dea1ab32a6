"""Bucketed, backward-overlapped gradient all-reduce for the Inception-v3 full fine-tune (SURVEY §2.7:
"Full Inception fine-tune ≈95 MB fp32", config 5 of BASELINE.md).

The trainer's gradient arena is ordered by readiness (head, then convs in reverse execution order), so the
gradients of everything *before* arena offset ``conv_end[name]`` are final once conv ``name``'s backward kernels
are enqueued.  ``ArenaBuckets`` cuts the arena into contiguous ranges of ≈``bucket_mb`` at conv boundaries;
``FinetuneDataParallel`` launches one asynchronous ``all_reduce`` per completed range from the trainer's
``on_conv_done`` hook — RCCL runs it on its own stream while the remaining backward GEMMs keep the CUs busy — and
waits for all of them right before the fused optimizer kernel, which folds in the ``1/world`` averaging.

Transport: on GPUs the buckets go through the framework's OWN communicator (``parallel/rccl_comm.py``,
``csrc/hip/rccl_comm.hip``: ``ncclAllReduce`` enqueued on a side stream that is ordered after the backward by events this
code owns — no ``torch.distributed`` collective inside the step, no process-group watchdog on the training streams).
``grad_dtype="bf16"`` casts each bucket to bf16 for the wire (44 MB instead of 87 MB per step) and back into the fp32
arena, where the optimizer keeps fp32 master weights.  gloo (CPU tests, several ranks on one GPU) keeps the process group.

Bucket size: xGMI rings are per-link bound (7 links x ≈153 GB/s), so a few ≈25 MB messages amortise the ring
latency better than the many 1-4 MB buckets tuned for NVSwitch; 87 MB of gradients -> 4 buckets by default.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class ArenaBuckets:
    def __init__(self, layout, bucket_mb: float = 25.0):
        limit = int(bucket_mb * 1e6 / 4)
        self.ranges: List[Tuple[int, int]] = []
        start = 0
        for name in layout.conv_order:
            end = layout.conv_end[name]
            if end - start >= limit:
                self.ranges.append((start, end))
                start = end
        if start < layout.size:
            self.ranges.append((start, layout.size))

    def ready(self, upto: int, cursor: int) -> int:
        """Index one past the last bucket that lies entirely below arena offset ``upto``."""
        k = cursor
        while k < len(self.ranges) and self.ranges[k][1] <= upto:
            k += 1
        return k


class FinetuneDataParallel:
    def __init__(self, trainer, *, bucket_mb: float = 25.0, group=None, overlap: bool = True,
                 force_collectives: bool = False, comm: str = "auto", grad_dtype: str = "fp32"):
        self.trainer = trainer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.overlap = overlap
        self._multi = self.world > 1 or (force_collectives and dist.is_initialized())  # 1-rank RCCL rehearsal
        self.buckets = ArenaBuckets(trainer.layout, bucket_mb)
        trainer.world = self.world
        # gloo cannot take device memory shared by several ranks of one GPU (functional tests): stage through the host
        self._stage_host = dist.is_initialized() and trainer.P.is_cuda and dist.get_backend(group) == "gloo"
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be fp32 or bf16")
        self.grad_dtype = grad_dtype
        self.rccl = None
        self.transport = "process-group"
        self.last_num_collectives = 0
        if self._multi and trainer.P.is_cuda and comm in ("auto", "rccl") and dist.get_backend(group) == "nccl":
            self._try_rccl(required=(comm == "rccl"))
        if self.grad_dtype == "bf16" and self.rccl is None:
            self.grad_dtype = "fp32"  # (the process-group path reduces the arena in place)
        if self._multi:
            self.broadcast_state()

    def _try_rccl(self, required: bool) -> None:
        """The framework's communicator (collective): every rank reports whether its part worked; all or none use it."""
        import os
        from .rccl_comm import RcclCommunicator
        tr = self.trainer
        ok, err, comm = 1, "", None
        try:
            comm = RcclCommunicator.create(self.rank, self.world, tr.P.device,
                                           key="rccl/ft%d" % int(os.environ.get("DTF_RCCL_GENERATION", "0")))
            x = torch.full((1024,), float(self.rank + 1), device=tr.P.device)
            comm.all_reduce(x)
            torch.cuda.synchronize(tr.P.device)
            if float(x[0].item()) != self.world * (self.world + 1) / 2.0:
                ok, err = 0, "native all-reduce returned %r" % float(x[0].item())
        except Exception as e:  # noqa: BLE001
            ok, err = 0, repr(e)
        flag = torch.tensor([ok], dtype=torch.int32, device=tr.P.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()):
            self.rccl = comm
            self.transport = "rccl-native"
            self._side = torch.cuda.Stream(device=tr.P.device)
            self._ev_ready = [torch.cuda.Event() for _ in self.buckets.ranges]
            self._ev_done = torch.cuda.Event()
            if self.grad_dtype == "bf16":
                n = max(b - a for a, b in self.buckets.ranges)
                self._wire = torch.empty((n + 7) // 8 * 8, dtype=torch.bfloat16, device=tr.P.device)
        else:
            if comm is not None:
                comm.close(abort=True)
            if required:
                raise RuntimeError("native RCCL communicator unavailable: %s" % (err or "a peer failed"))

    def _all_reduce(self, t: torch.Tensor, async_op: bool):
        if not self._stage_host:
            return dist.all_reduce(t, group=self.group, async_op=async_op)
        h = t.cpu()
        dist.all_reduce(h, group=self.group)
        t.copy_(h)
        return None

    def broadcast_state(self, src: int = 0) -> None:
        tensors = [self.trainer.P]
        for extra in ("M", "V", "moving_mean", "moving_var"):
            if hasattr(self.trainer, extra):
                tensors.append(getattr(self.trainer, extra))
        for t in tensors:
            if self.rccl is not None:
                self.rccl.broadcast(t, root=src)
            elif self._stage_host:
                h = t.cpu()
                dist.broadcast(h, src=src, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=self.group)

    # ------------------------------------------------------------------ the step
    def _reduce_bucket_native(self, k: int) -> None:
        """Bucket ``k`` of the gradient arena is final on the compute stream: sum it over the ranks on the side stream."""
        from .. import native
        tr = self.trainer
        a, b = self.buckets.ranges[k]
        main = torch.cuda.current_stream(tr.P.device)
        self._ev_ready[k].record(main)
        self._side.wait_event(self._ev_ready[k])
        g = tr.G[a:b]
        if self.grad_dtype == "bf16" and (b - a) % 8 == 0:
            lib, st = native.hip(), self._side.cuda_stream
            wire = self._wire[:b - a]
            native.check(lib.dtf_cast_f32_bf16(g.data_ptr(), wire.data_ptr(), b - a, st), "cast_f32_bf16")
            self.rccl.all_reduce(wire, stream=self._side)
            native.check(lib.dtf_cast_bf16_f32(wire.data_ptr(), g.data_ptr(), b - a, st), "cast_bf16_f32")
        else:
            self.rccl.all_reduce(g, stream=self._side)

    def fwd_bwd_exchange(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        """Forward, backward and the gradient exchange; on return (stream order) ``G`` holds the sum over the ranks."""
        tr = self.trainer
        cursor = [0]
        works = []

        def launch(upto_bucket: int) -> None:
            while cursor[0] < upto_bucket:
                if self.rccl is not None:
                    self._reduce_bucket_native(cursor[0])
                else:
                    a, b = self.buckets.ranges[cursor[0]]
                    w = self._all_reduce(tr.G[a:b], async_op=True)
                    if w is not None:
                        works.append(w)
                cursor[0] += 1

        def on_conv_done(name: str) -> None:
            if self.overlap:
                launch(self.buckets.ready(tr.layout.conv_end[name], cursor[0]))

        tr.fwd_bwd(x, labels, on_conv_done=on_conv_done)
        launch(len(self.buckets.ranges))
        if self.rccl is not None:
            self._ev_done.record(self._side)
            torch.cuda.current_stream(tr.P.device).wait_event(self._ev_done)
        for w in works:
            w.wait()
        self.last_num_collectives = cursor[0]

    def graph_capable(self) -> bool:
        """The whole step is kernels and stream-ordered communicator calls (capturable into a hipGraph): the native trainer
        on one GPU, or with the framework's RCCL communicator.  The process-group path waits for work handles on the host."""
        tr = self.trainer
        return bool(tr.P.is_cuda and hasattr(tr, "_fwd_parity") and (not self._multi or self.rccl is not None))

    def train_step(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        tr = self.trainer
        if not self._multi:
            tr.train_step(x, labels)
            return
        self.fwd_bwd_exchange(x, labels)
        tr.apply()

    def check_exchange(self, x: torch.Tensor, labels: torch.Tensor, tol: Optional[float] = None) -> float:
        """Debug assertion on the exchange (the fine-tune's ``check_exchange``), in ONE backward pass (two
        passes of this network are not comparable number by number: atomically accumulated filter gradients and
        batch-norm statistics round differently from run to run and 94 layers amplify it):

        1. readiness: the backward runs with the production hook, but a bucket that the hook declares ready is only
           SNAPSHOT at that point of the stream; after the backward every snapshot must equal the final local gradient bit
           for bit — a bucket launched before its last producer would differ;
        2. transport: the local gradient is summed over the ranks by the process group, host-ordered (the reference), and
           by the production path (native communicator on the side stream, bucket by bucket, optional bf16 wire).

        Returns the relative deviation of 2 (max over ranks), raises beyond ``tol`` or on a readiness violation; no step is
        applied and the gradient arena is left zeroed.  Collective."""
        tr = self.trainer
        if not self._multi:
            return 0.0
        if tol is None:
            tol = 2e-2 if self.grad_dtype == "bf16" else 1e-5
        tr.G.zero_()
        snap = torch.empty_like(tr.G)
        cursor = [0]

        def take(upto_bucket: int) -> None:
            while cursor[0] < upto_bucket:
                a, b = self.buckets.ranges[cursor[0]]
                snap[a:b].copy_(tr.G[a:b])
                cursor[0] += 1

        tr.fwd_bwd(x, labels, on_conv_done=lambda name: take(self.buckets.ready(tr.layout.conv_end[name], cursor[0])))
        take(len(self.buckets.ranges))
        early = int((snap != tr.G).sum().item())
        ref = tr.G.clone()
        if self._stage_host or not ref.is_cuda:
            h = ref.cpu()
            dist.all_reduce(h, group=self.group)
            ref = h.to(ref.device)
        else:
            dist.all_reduce(ref, group=self.group)
        # the production transport on the same local gradient
        works = []
        for k, (a, b) in enumerate(self.buckets.ranges):
            if self.rccl is not None:
                self._reduce_bucket_native(k)
            else:
                w = self._all_reduce(tr.G[a:b], async_op=True)
                if w is not None:
                    works.append(w)
        if self.rccl is not None:
            self._ev_done.record(self._side)
            torch.cuda.current_stream(tr.P.device).wait_event(self._ev_done)
        for w in works:
            w.wait()
        dev = ((tr.G - ref).norm() / (ref.norm() + 1e-30)).double().reshape(1)
        tr.G.zero_()
        t = torch.cat([dev.cpu(), torch.tensor([float(early)], dtype=torch.float64)])
        if not (self._stage_host or not tr.G.is_cuda):
            t = t.to(tr.G.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        d, early = float(t[0].item()), int(t[1].item())
        if early:
            raise AssertionError("fine-tune gradient exchange: %d gradient elements changed after their bucket had been "
                                 "declared ready (a bucket would be reduced before its last producer)" % early)
        if not d < tol:
            raise AssertionError("fine-tune gradient exchange (%s, %s wire): deviates %.3g from the host-ordered sum of the "
                                 "ranks' local gradients" % (self.transport, self.grad_dtype, d))
        return d

    def async_error(self) -> Optional[str]:
        """The native communicator's asynchronous error (a peer died, a transport failed), polled at logging points."""
        return self.rccl.async_error() if self.rccl is not None else None

    def close(self) -> None:
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None

    def replicas_in_sync(self) -> bool:
        if self.world == 1:
            return True
        p = self.trainer.P
        cs = torch.stack([p.double().sum(), p.double().abs().sum()]).cpu()
        lo, hi = cs.clone(), cs.clone()
        if not self._stage_host and p.is_cuda:
            lo, hi = lo.to(p.device), hi.to(p.device)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo.cpu(), hi.cpu()))

    def mean_scalar(self, v: float) -> float:
        if self.world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64)
        if not self._stage_host and self.trainer.P.is_cuda and dist.get_backend(self.group) != "gloo":
            t = t.to(self.trainer.P.device)
        dist.all_reduce(t, group=self.group)
        return float(t.item()) / self.world

    def mean_metrics(self) -> Tuple[float, float]:
        loss, acc = self.trainer.metrics()
        if self.world == 1:
            return loss, acc
        t = torch.tensor([loss, acc], dtype=torch.float64)
        if not self._stage_host and self.trainer.P.is_cuda and dist.get_backend(self.group) != "gloo":
            t = t.to(self.trainer.P.device)
        dist.all_reduce(t, group=self.group)
        return tuple((t / self.world).cpu().tolist())
