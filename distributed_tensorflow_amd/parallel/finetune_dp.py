"""Bucketed, backward-overlapped gradient all-reduce for the Inception-v3 full fine-tune (SURVEY §2.7:
"Full Inception fine-tune ≈95 MB fp32", config 5 of BASELINE.md).

The trainer's gradient arena is ordered by readiness (head, then convs in reverse execution order), so the
gradients of everything *before* arena offset ``conv_end[name]`` are final once conv ``name``'s backward kernels
are enqueued.  ``ArenaBuckets`` cuts the arena into contiguous ranges of ≈``bucket_mb`` at conv boundaries;
``FinetuneDataParallel`` launches one asynchronous ``all_reduce`` per completed range from the trainer's
``on_conv_done`` hook — RCCL runs it on its own stream while the remaining backward GEMMs keep the CUs busy — and
waits for all of them right before the fused optimizer kernel, which folds in the ``1/world`` averaging.

Bucket size: xGMI rings are per-link bound (7 links x ≈153 GB/s), so a few ≈25 MB messages amortise the ring
latency better than the many 1-4 MB buckets tuned for NVSwitch; 87 MB of gradients -> 4 buckets by default.
"""
from __future__ import annotations

from typing import List, Tuple

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
                 force_collectives: bool = False):
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
        if self._multi:
            self.broadcast_state()

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
            if self._stage_host:
                h = t.cpu()
                dist.broadcast(h, src=src, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src, group=self.group)

    def train_step(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        tr = self.trainer
        if not self._multi:
            tr.train_step(x, labels)
            return
        works = []
        cursor = [0]

        def launch(upto_bucket: int) -> None:
            while cursor[0] < upto_bucket:
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
        for w in works:
            w.wait()
        self.last_num_collectives = cursor[0]
        tr.apply()

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

    def mean_metrics(self) -> Tuple[float, float]:
        loss, acc = self.trainer.metrics()
        if self.world == 1:
            return loss, acc
        t = torch.tensor([loss, acc], dtype=torch.float64)
        if not self._stage_host and self.trainer.P.is_cuda and dist.get_backend(self.group) != "gloo":
            t = t.to(self.trainer.P.device)
        dist.all_reduce(t, group=self.group)
        return tuple((t / self.world).cpu().tolist())
