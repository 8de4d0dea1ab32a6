"""Synchronous data parallelism over RCCL/xGMI (component C22 replacement).

The reference's only parallel strategy is asynchronous parameter-server replication over gRPC:
every worker pulls 13.1 MB of variables and pushes 13.1 MB of gradients per step through one PS
(``demo2/train.py:27-29,146-149,181-184``).  Here every GPU holds a full replica and
gradients are summed with ``torch.distributed.all_reduce`` (backend "nccl" == RCCL on ROCm;
"gloo" for CPU tests) directly on slices of the engine's flat gradient arena:

* two buckets in gradient-readiness order — A = fc1/fc2 (12.9 MB, ready after step part 1) and
  B = conv1/conv2 (0.2 MB, ready after part 2) — so the big all-reduce runs on RCCL's stream while
  the conv backward is still executing (xGMI is point-to-point: few large messages beat many small);
* the 1/world averaging is folded into the fused Adam kernel (``grad_scale``), no extra pass;
* rank 0's parameters (and optimizer state on resume) are broadcast once at start — the
  replacement for "chief runs init_op, others wait" (``demo2/train.py:166-176``);
* metrics (loss sum, correct, count) and the stop flag are reduced only at logging points.

``--async`` emulation (reference semantics, quirk Q6): ranks apply their local gradients every
step and parameters are averaged every ``async_period`` steps (local SGD).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

from ..utils import trace


def init_process_group(backend: str = "auto", rank: Optional[int] = None, world: Optional[int] = None,
                       master_addr: Optional[str] = None, master_port: Optional[int] = None,
                       timeout_s: int = 600) -> Tuple[int, int]:
    """Initialise torch.distributed from explicit arguments or the torchrun environment."""
    import datetime

    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
    if world <= 1:
        return 0, 1
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    ndev = torch.cuda.device_count()
    if backend == "auto":
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if backend == "nccl" and ndev < local_world:
            # more ranks on this host than GPUs (functional runs, rehearsals on a one-GPU box): RCCL refuses two ranks on
            # one device, the ranks share the GPUs and the process group is gloo (device tensors staged, see all_reduce_)
            backend = "gloo"
            if rank == 0:
                print("%d ranks on this host for %d GPU(s): ranks share devices, process group backend gloo" % (local_world, ndev),
                      flush=True)
    if master_addr is not None:
        os.environ["MASTER_ADDR"] = master_addr
    if master_port is not None:
        os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev > 0 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % ndev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=timeout_s))
    return rank, world


def _needs_host_staging(t: torch.Tensor, group=None) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t: torch.Tensor, group=None, op=None) -> None:
    """In-place all-reduce that also works for device tensors on a gloo group (functional runs with several ranks on
    one GPU, where RCCL refuses duplicate devices): those are staged through host memory explicitly — gloo's own CUDA
    path was seen to stall after a few dozen calls with two ranks on one device."""
    op = dist.ReduceOp.SUM if op is None else op
    if _needs_host_staging(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def broadcast_(t: torch.Tensor, src: int = 0, group=None) -> None:
    """In-place broadcast with the same host staging rule as :func:`all_reduce_`."""
    if _needs_host_staging(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


from .dp_startup import StartupMixin
from .grad_hooks import IpcGradHook as _IpcGradHook, RcclGradHook as _RcclGradHook  # noqa: F401  (names kept for importers)
from .sharded_state import ShardedStateMixin


class DataParallelTrainer(StartupMixin, ShardedStateMixin):
    """Drives an engine (native or eager) under synchronous or emulated-async data parallelism."""

    def __init__(self, engine, *, sync: bool = True, async_period: int = 8, group=None, overlap: bool = True,
                 force_collectives: bool = False, comm: str = "auto"):
        self.engine = engine
        self._wsh = False          # the "ipc-wshard" schedule is active (see _wsh_enter)
        # force_collectives: take the multi-rank code path even with one rank (lets a 1-GPU box exercise the
        # RCCL stream/event choreography of the overlapped step end to end)
        self._force = force_collectives and dist.is_initialized()
        self.group = group
        self.sync = sync
        self.async_period = max(1, async_period)
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        engine.set_world(self.world if sync else 1)
        (a0, a1), (b0, b1) = engine.bucket_bounds()
        self._bucket_a = engine.G[a0:a1]
        self._bucket_b = engine.G[b0:b1]
        self._native = hasattr(engine, "stream")
        # gloo with device tensors (functional tests: several ranks sharing one GPU, where RCCL refuses duplicate
        # devices) is staged through host memory explicitly; RCCL reduces the arena slices in place.
        self._stage_host = (dist.is_initialized() and self._native and dist.get_backend(group) == "gloo")
        self._multi = self.world > 1 or self._force
        self._pg_stream = None
        if self._multi:
            self.broadcast_state()
        # gradient transport: "ipc" = direct two-shot all-reduce over IPC-mapped peer memory (all xGMI links at once, and
        # plain kernels on the engine's stream, so the whole step is one hipGraph), "rccl" = torch.distributed
        # collectives overlapped with the conv backward, "auto" = ipc when every rank is on this host, the engine is
        # native and the start-up self-test against the process group's own all-reduce passes, else rccl.
        self.comm = "rccl"
        self.ipc = None
        self.rccl = None           # native RCCL communicator (collectives captured into the step graph)
        self._rccl_graph = False   # the current schedule is that one
        self._pg_stream = None
        self.transport_us = None
        if comm not in ("auto", "ipc", "rccl"):
            raise ValueError("comm must be auto, ipc or rccl")
        self.schedule_us = None
        self.schedule = None
        self.ipc_act = None
        self.ipc_w = None          # landing buffers of the shard schedule (fc1 filter gradient, 12.8 MB per rank)
        # "ipc-wshard" schedule: W_fc1 lives in ipc_w's buffers and its optimizer state is sharded over the ranks; the live
        # arenas are then not a consistent state between two materialize() calls (see there)
        self._wsh_snap = None      # last consistent (P, M, V, step) on this rank, what a checkpoint is written from
        self._wsh_every = max(16, int(os.environ.get("DTF_WSHARD_SNAPSHOT_STEPS", "2000")))
        self._shared_gpu = False
        import time
        t_start = self._t_start = time.perf_counter()
        self.startup_log = []
        if self._multi and self._native and sync and not self._stage_host and comm in ("auto", "rccl"):
            self._try_rccl()
        if self._multi and self._native and sync and comm in ("auto", "ipc"):
            t_a = time.perf_counter()
            self._try_ipc(required=(comm == "ipc"))
            t_b = time.perf_counter()
            if self.comm == "ipc":
                self._tune_schedule(allow_rccl=(comm == "auto"))
            self._note("start-up phases: RCCL communicator %.2f s, IPC mapping + self-test + transport timing %.2f s, "
                       "exchange buffers + schedule trials %.2f s" % (t_a - t_start, t_b - t_a, time.perf_counter() - t_b))
        if self._multi and self._native and sync and self.comm != "ipc" and self.rccl is not None:
            self._ipc_hook, self._fc1_end = None, 0
            self._use_schedule("rccl")
            if not self._all_ranks_ok(self._capture_ok()):
                # RCCL calls could not be captured into the step graph on some rank: every rank falls back to the
                # process group's collectives between three graph launches
                self._note("the RCCL collectives could not be captured into the step graph: process-group schedule instead")
                self.engine.set_grad_hook(None)
                self._rccl_graph = False
                self.rccl.close(abort=True)
                self.rccl = None
        self.startup_seconds = round(time.perf_counter() - t_start, 3) if self._multi else 0.0
        if self._multi and self.rank == 0:
            print("data-parallel start-up: %.2f s; transport %s, schedule %s%s" % (
                self.startup_seconds,
                self.comm if self.comm != "rccl" else "process group (%s)" % dist.get_backend(self.group), self.schedule,
                "".join("\n  " + l for l in self.startup_log)), flush=True)

    def ensure_healthy(self) -> bool:
        """Call between steps (synchronises).  If an in-kernel wait of the IPC exchange has expired on any rank since
        the start, the ranks agree on it, fall back to the RCCL schedule and re-broadcast rank 0's state (some step
        used an incomplete gradient sum on some rank).  Returns False if it had to do that."""
        if self.ipc is None or self.comm != "ipc":
            return True
        e = self.engine
        t = torch.tensor([float(self.transport_errors())], dtype=torch.float64, device="cpu" if self._stage_host else e.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        if float(t.item()) == 0:
            return True
        if self.rank == 0:
            print("IPC exchange: %d wait(s) timed out; falling back to the process group's all-reduce" % int(t.item()),
                  flush=True)
        self._use_schedule("rccl")
        self.broadcast_state()  # (also realigns the host's step mirror: a skipped step did not advance the device counter)
        return False

    # ------------------------------------------------------------------ collectives
    def _ctx(self):
        """Context for process-group collectives and the torch ops around them: they run on a stream of their own that is
        ordered after everything enqueued on the engine's stream so far, and the engine's stream continues after them.
        ROOT CAUSE this removes: the process group records its own synchronisation events on whatever stream is current
        when a collective is issued, and its watchdog thread keeps querying them; issued on the engine's stream, such an
        event could still be held by the watchdog when that stream went into graph capture, and the query then failed
        with hipErrorCapturedEvent and aborted the process (once in a few dozen runs; round 1 papered over it with a
        5 ms sleep).  The engine's stream now only ever carries events this code owns."""
        import contextlib
        if not self._native:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def isolated():
            e = self.engine
            if self._pg_stream is None:
                self._pg_stream = torch.cuda.Stream(device=e.device)
            self._pg_stream.wait_stream(e.stream)
            try:
                with torch.cuda.stream(self._pg_stream):
                    yield
            finally:  # (also when the body raised: the engine's stream must not run ahead of a half-issued collective)
                e.stream.wait_stream(self._pg_stream)
        return isolated()

    def _reduce_async(self, t: torch.Tensor):
        """Asynchronous process-group all-reduce of ``t``, ordered after the engine's stream, issued on the collective
        stream (see ``_ctx``); returns the work handle (None when the backend is host-staged and therefore synchronous)."""
        e = self.engine
        if self._stage_host:
            with torch.cuda.stream(e.stream):
                return self._reduce(t)
        if self._pg_stream is None:
            self._pg_stream = torch.cuda.Stream(device=e.device)
        self._pg_stream.wait_stream(e.stream)
        with torch.cuda.stream(self._pg_stream):
            return dist.all_reduce(t, group=self.group, async_op=True)

    def _reduce(self, t: torch.Tensor, op=None, async_op: bool = False):
        """all-reduce ``t`` in place on the engine's stream (host-staged when the backend cannot take device memory)."""
        op = dist.ReduceOp.SUM if op is None else op
        if not self._stage_host:
            return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)
        h = t.to("cpu")  # stream-ordered D2H + wait
        dist.all_reduce(h, op=op, group=self.group)
        t.copy_(h)
        return None

    def _bcast(self, t: torch.Tensor, src: int) -> None:
        if not self._stage_host:
            dist.broadcast(t, src=src, group=self.group)
            return
        h = t.to("cpu")
        dist.broadcast(h, src=src, group=self.group)
        t.copy_(h)

    def broadcast_state(self, src: int = 0) -> None:
        """rank-``src`` parameters + optimizer state + step -> every replica."""
        e = self.engine  # (rank src's ARENAS are what counts: after materialize(), or after a restore)
        with self._ctx():
            for t in (e.P, e.M, e.V):
                self._bcast(t, src)
            step = e.step_t if self._native else torch.tensor([e.global_step], dtype=torch.int32)
            self._bcast(step, src) if self._native else dist.broadcast(step, src=src, group=self.group)
        if self._native:
            e.stream.synchronize()
            e._steps_done = int(e.step_t.item())
        else:
            e._steps_done = int(step.item())
        self._wsh_reload()

    def close(self) -> None:
        """Release the IPC exchange buffers (collective: every rank calls it, before the process group is destroyed)."""
        self._wsh_leave()
        if self.ipc is not None:
            self.engine.set_grad_redirect(0, 0)
            if hasattr(self.engine, "set_shard"):
                self.engine.set_shard(None)
            elif hasattr(self.engine, "set_gather"):
                self.engine.set_gather(None)
            self.engine.set_grad_hook(None)
            for name in ("ipc_w", "ipc_act"):
                if getattr(self, name) is not None:
                    getattr(self, name).close()
                    setattr(self, name, None)
            self.ipc.close()
            self.ipc = None
            self.comm = "rccl"
        if self.rccl is not None:
            if self._rccl_graph:
                self.engine.set_grad_hook(None)  # (drops the graphs that hold the communicator's kernels)
                self._rccl_graph = False
            self.rccl.close()
            self.rccl = None

    def transport_errors(self) -> int:
        """Expired in-kernel waits of the IPC exchanges so far (synchronises); 0 when IPC is not in use."""
        return sum(c.errors() for c in (self.ipc, self.ipc_act, self.ipc_w) if c is not None)

    def check_transport(self) -> None:
        """Raise if an in-kernel wait of the IPC all-reduce ever expired (a peer stalled for longer than the bound, so
        some step used an incomplete gradient sum).  Called at logging points; synchronises."""
        if self.ipc is not None:
            n = self.transport_errors()
            if n:
                raise RuntimeError("IPC gradient all-reduce: %d wait(s) timed out on rank %d; a peer stalled or died" %
                                   (n, self.rank))

    def replicas_in_sync(self) -> bool:
        """Debug invariant: parameter checksums identical on every rank."""
        if not self._multi:
            return True
        e = self.engine
        self.sync_params()
        with self._ctx():
            cs = torch.stack([e.P.double().sum(), e.P.double().abs().sum()])
            lo, hi = cs.clone(), cs.clone()
            if self._stage_host:
                lo, hi = lo.cpu(), hi.cpu()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo.cpu(), hi.cpu()))

    def check_exchange(self, tol: float = 2e-3) -> float:
        """Debug assertion on the gradient exchange itself (``--check_replicas``).  ONE checked step:

        1. this rank's local gradient of the coming step is computed with no exchange at all and summed over the ranks by
           the process group, host-ordered on its own stream (the reference);
        2. the step then runs through the production schedule (IPC kernels / RCCL inside the step graph / bucketed
           process-group calls), and the gradient the optimizer actually consumed is recovered from Adam's first moment,
           ``m' = b1 m + (1 - b1) g``.

        Both passes draw the same batch and the same dropout mask (keyed by the step counter).  A stream-ordering bug —
        the optimizer running before the exchange has landed, a bucket reduced before its producer finished — shows as a
        deviation here long before the loss curve notices.  Returns the relative deviation (max over ranks); raises
        beyond ``tol``.  Collective; advances training by exactly one (production) step."""
        if not (self._multi and self.sync):
            return 0.0
        e = self.engine
        if getattr(e, "dataset", None) is not None and e.spe - (e.global_step % e.spe) < 2:
            self.train_steps(1)  # (not across the reshuffle at an epoch boundary)
        b1 = float(e.cfg.beta1) if hasattr(e, "cfg") else float(getattr(e, "b1", 0.9))
        self.materialize()  # (wshard: the local pass reads P, the comparison every rank's rows of the first moment)
        if self._native:
            cfg, keep = e.cfg, (e.cfg.G_fc1_dst, e.cfg.G_src, e.cfg.gather_n, e.cfg.shard_n, e.cfg.wshard)
            with self._ctx():
                ring = e.metrics_ring.clone()
            cfg.G_fc1_dst, cfg.G_src, cfg.gather_n, cfg.shard_n, cfg.wshard = None, None, 0, 0, 0   # plain local gradients
            try:                                  # into G (graphs are untouched: their arguments were bound at capture)
                with torch.cuda.stream(e.stream):
                    e.G.zero_()
                    e.fwd_bwd(3)
            finally:
                cfg.G_fc1_dst, cfg.G_src, cfg.gather_n, cfg.shard_n, cfg.wshard = keep
            with self._ctx():
                ref = e.G.clone()
                e.G.zero_()
                e.metrics_ring.copy_(ring)  # (the local pass added this step's loss / accuracy once already)
                self._reduce(ref)
                m0 = e.M.clone()
        else:
            gen = e._gen.get_state()
            ring = e._ring.copy()
            e.fwd_bwd(3)
            e._gen.set_state(gen)
            e._ring[:] = ring
            ref = e.G.clone()
            e.G.zero_()
            dist.all_reduce(ref, group=self.group)
            m0 = e.M.clone()
        scale = 1.0 / self.world  # (local gradients are means over the local batch; the optimizer divides the sum by world)
        self.train_steps(1)
        self.materialize()
        with self._ctx():
            # compared in first-moment space (recovering g itself cancels catastrophically once |g| << |m|); the floor
            # keeps the rounding of m' (~6e-8 |m'|) below the tolerance when the gradient has all but vanished
            want = b1 * m0 + (1.0 - b1) * scale * ref
            dev = (e.M - want).norm() / ((1.0 - b1) * scale * ref.norm() + 1e-4 * e.M.norm() + 1e-30)
            t = dev.double().reshape(1).to("cpu" if self._stage_host else e.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        dev = float(t.item())
        if not dev < tol:
            raise AssertionError("gradient exchange (%s): the optimizer consumed a gradient that deviates %.3g from the sum "
                                 "of the ranks' local gradients" % (self.schedule or self.comm, dev))
        return dev

    def all_reduce_scalars(self, values) -> list:
        """Sum a few python floats across ranks (logging points only)."""
        if self.world == 1:
            return [float(v) for v in values]
        self.check_transport()
        dev = "cpu" if self._stage_host else self.engine.device
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        with self._ctx():
            dist.all_reduce(t, group=self.group)
        return t.cpu().tolist()

    def any_rank(self, flag: bool) -> bool:
        """all-reduce(max) of a stop flag."""
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if self._stage_host else self.engine.device)
        with self._ctx():
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(int(t.item()))

    def barrier(self) -> None:
        if self.world > 1:
            with self._ctx():
                dist.barrier(group=self.group)

    # ------------------------------------------------------------------ stepping
    def prepare(self) -> None:
        """Capture the step graphs ahead of the first step (collective-free; every rank may call it)."""
        if self._native and (not self._multi or (self.sync and self._one_graph())):
            self.engine.prepare()
        elif self._native and self._multi and self.sync:
            self._capture_rccl_graphs()

    def _one_graph(self) -> bool:
        """The whole data-parallel step (gradient exchange included) is kernels / RCCL calls on the engine's streams."""
        return self.comm == "ipc" or self._rccl_graph

    def _capture_rccl_graphs(self) -> None:
        """The process-group schedule (no native communicator: gloo, or RCCL that could not be bound) replays three graphs
        per step with ``torch.distributed`` collectives in between; all three are captured up front.
        The process group's collectives never run on the engine's stream (see ``_pg``), which is what used to end,
        once in a few dozen runs, in ``hipErrorCapturedEvent`` inside the process group's watchdog thread."""
        e = self.engine
        if not getattr(e, "use_graph", False):
            return
        for k in ("part1", "part2", "apply"):
            if k not in e._graphs:
                e._capture(k)

    def train_steps(self, n: int) -> None:
        """``n`` synchronous steps; several per graph launch where the whole step (incl. the gradient all-reduce) is
        plain kernels on the engine's stream (single replica, or the IPC transport)."""
        e = self.engine
        if self._wsh:
            # sharded optimizer state: every _wsh_every steps (the same steps on every rank) the ranks assemble the
            # consistent state that checkpoints are written from
            while n > 0:
                k = min(n, self._wsh_every - e.global_step % self._wsh_every)
                self._wsh_dirty = True
                e.train_steps(k)
                n -= k
                if e.global_step % self._wsh_every == 0:
                    self.materialize(snapshot=True)
            return
        if self._native and (not self._multi or (self.sync and self._one_graph())):
            e.train_steps(n)
            return
        for _ in range(n):
            self.train_step()

    def train_step(self) -> None:
        e = self.engine
        if self._wsh:
            self.train_steps(1)
            return
        if not self._multi:
            e.train_step()
            return
        if not self.sync:
            e.train_step()
            if e.global_step % self.async_period == 0:
                with self._ctx():
                    self._reduce(e.P)
                    e.P.div_(self.world)
            return
        if self._native and self._one_graph():
            with trace.range("fwd+bwd | %s all-reduce | Adam (one graph)" % self.comm.upper()):
                e.launch("full")
            return
        if self._native:
            self._capture_rccl_graphs()
            with torch.cuda.stream(e.stream):
                with trace.range("fwd+head+fc_dW (graph part1)"):
                    e.launch("part1")
                if self.overlap:
                    with trace.range("all-reduce bucket A (fc grads) || conv backward (graph part2)"):
                        wa = self._reduce_async(self._bucket_a)
                        e.launch("part2")
                    with trace.range("all-reduce bucket B (conv grads)"):
                        wb = self._reduce_async(self._bucket_b)
                        if wa is not None:
                            wa.wait()  # (the engine's stream waits for the collective's end event)
                            wb.wait()
                else:
                    with trace.range("conv backward (graph part2) + all-reduce"):
                        e.launch("part2")
                        w = self._reduce_async(e.G)
                        if w is not None:
                            w.wait()
                with trace.range("fused Adam (graph apply)"):
                    e.launch("apply")
        else:
            e.fwd_bwd(3)
            dist.all_reduce(e.G, group=self.group)
            e.apply()
