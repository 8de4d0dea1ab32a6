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
    if backend == "auto":
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if master_addr is not None:
        os.environ["MASTER_ADDR"] = master_addr
    if master_port is not None:
        os.environ["MASTER_PORT"] = str(master_port)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
        torch.cuda.set_device(local)
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


class _IpcGradHook:
    """Gradient all-reduce of the native MNIST step over the IPC transport.  With ``overlap`` the fc bucket (12.9 MB,
    final after step part 1) is reduced on a forked stream while the conv backward (part 2) runs on the main one —
    event fork/join, so it also becomes a parallel branch of the captured hipGraph — and only the 0.2 MB conv bucket
    is reduced after it."""

    def __init__(self, ipc, bucket_a: torch.Tensor, bucket_b: torch.Tensor, whole: torch.Tensor, overlap: bool):
        self.ipc, self.a, self.b, self.whole, self.overlap = ipc, bucket_a, bucket_b, whole, overlap
        # zero-copy mode (set_zero_copy): the fc1 filter gradient, arena [0, zc_off), is produced straight into the
        # exchange buffer and the optimizer reads the reduced arena from the result buffer; see ipc_allreduce.hip
        self.zc_off = 0
        # gather mode (set_gather): the fc1 filter gradient, arena [0, rest_off), is computed globally from gathered
        # operands inside the step; only the rest of the arena is all-reduced
        self.rest_off = 0
        # (the engine's DTF_EARLY_FC1_ADAM=0 debugging switch applies the whole arena in one optimizer launch, which then
        # needs the rest of the arena reduced in place as before)
        self.fused_apply = os.environ.get("DTF_EARLY_FC1_ADAM", "1") != "0"
        if overlap:
            self.side = torch.cuda.Stream(device=whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def set_overlap(self, overlap: bool) -> None:
        self.overlap = overlap
        if overlap and not hasattr(self, "side"):
            self.side = torch.cuda.Stream(device=self.whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def set_zero_copy(self, rest_off: int) -> None:
        self.zc_off = rest_off
        if rest_off:
            self.overlap = False

    def set_gather(self, rest_off: int) -> None:
        self.rest_off = rest_off
        if rest_off:
            self.overlap = False

    @property
    def apply_comm(self):
        """Non-None in gather mode: the engine's optimizer launch for the arena after W_fc1 must sum the ranks' published
        gradients itself (``dtf_mnist_apply_part_ipc`` on this comm struct)."""
        return self.ipc.comm if self.rest_off and self.fused_apply else None

    def after_part1(self, stream) -> None:
        if not self.overlap:
            return
        self.ev_fork.record(stream)
        self.side.wait_event(self.ev_fork)
        self.ipc.all_reduce(self.a, stream=self.side.cuda_stream)
        self.ev_join.record(self.side)

    def after_part2(self, stream) -> None:
        if self.rest_off:
            # one-shot form (its "next call only after another barrier" contract is met by the operand exchange's
            # barrier after the next step's loss head), with the reduction fused into the optimizer launch: only the
            # copy into the exchange buffer happens here, see apply_comm
            if self.fused_apply:
                self.ipc.publish(self.whole[self.rest_off:], stream=stream.cuda_stream)
            else:
                self.ipc.all_reduce_oneshot(self.whole[self.rest_off:], stream=stream.cuda_stream)
            return
        if self.zc_off:
            self.ipc.all_reduce_zero_copy(self.whole[self.zc_off:], self.zc_off, self.whole.numel(),
                                          stream=stream.cuda_stream)
            return
        if not self.overlap:
            self.ipc.all_reduce(self.whole, stream=stream.cuda_stream)
            return
        stream.wait_event(self.ev_join)
        self.ipc.all_reduce(self.b, stream=stream.cuda_stream)


class _RcclGradHook:
    """Gradient all-reduce of the native MNIST step over the native RCCL communicator (``parallel/rccl_comm.py``): the
    ncclAllReduce calls are enqueued on the engine's stream(s) inside the step, so they are captured into the step's
    hipGraph and the whole data-parallel step is one graph launch.  With ``overlap`` the fc bucket (12.9 MB, final after
    step part 1) is reduced on a forked stream while the conv backward runs on the main one, as in ``_IpcGradHook``."""

    apply_comm = None

    def __init__(self, rccl, bucket_a: torch.Tensor, bucket_b: torch.Tensor, whole: torch.Tensor, overlap: bool):
        self.rccl, self.a, self.b, self.whole, self.overlap = rccl, bucket_a, bucket_b, whole, overlap
        if overlap:
            self.side = torch.cuda.Stream(device=whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def after_part1(self, stream) -> None:
        if not self.overlap:
            return
        self.ev_fork.record(stream)
        self.side.wait_event(self.ev_fork)
        self.rccl.all_reduce(self.a, stream=self.side)
        self.ev_join.record(self.side)

    def after_part2(self, stream) -> None:
        if not self.overlap:
            self.rccl.all_reduce(self.whole, stream=stream)
            return
        stream.wait_event(self.ev_join)
        self.rccl.all_reduce(self.b, stream=stream)


class DataParallelTrainer:
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
                self.startup_seconds, self.comm, self.schedule,
                "".join("\n  " + l for l in self.startup_log)), flush=True)

    def _note(self, msg: str) -> None:
        self.startup_log.append(msg)

    def _capture_ok(self) -> bool:
        """Capture this schedule's step graphs now; False (with a note) instead of an exception if that fails."""
        try:
            self.prepare()
            return True
        except Exception as e:  # noqa: BLE001
            self._note("%s: graph capture failed on rank %d: %r" % (self.schedule, self.rank, e))
            return False

    def _all_ranks_ok(self, ok: bool) -> bool:
        """Agreement (process group, collective stream): did every rank succeed?"""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device="cpu" if self._stage_host else self.engine.device)
        with self._ctx():  # (the read is ordered on the collective stream too: .item() outside would use the caller's)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            ok_all = float(t.item()) > 0
        return ok_all

    def _try_rccl(self) -> None:
        """Create the native RCCL communicator (collective).  Every rank reports whether its part worked; if any failed
        all of them drop it and the process group's own collectives are used instead."""
        import time
        t0 = time.perf_counter()
        from .rccl_comm import RcclCommunicator
        ok, err, comm = 1, "", None
        if dist.get_backend(self.group) != "nccl":
            return
        try:
            comm = RcclCommunicator.create(self.rank, self.world, self.engine.device,
                                           key="rccl/%d" % int(os.environ.get("DTF_RCCL_GENERATION", "0")))
            with self._ctx():
                x = torch.full((1024,), float(self.rank + 1), device=self.engine.device)
                comm.all_reduce(x)
            self.engine.stream.synchronize()
            if float(x[0].item()) != self.world * (self.world + 1) / 2.0:
                ok, err = 0, "native all-reduce returned %r" % float(x[0].item())
        except Exception as e:  # noqa: BLE001
            ok, err = 0, repr(e)
        with self._ctx():
            flag = torch.tensor([ok], dtype=torch.int32, device=self.engine.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            agreed = int(flag.item())
        if agreed:
            self.rccl = comm
            self._note("native RCCL communicator: %d ranks, library %s, %.2f s" % (self.world, comm.version, time.perf_counter() - t0))
        else:
            if comm is not None:
                comm.close(abort=True)
            self._note("native RCCL communicator not used (%s)" % (err or "a peer failed"))

    def _time_transports(self, iters: int = 8, grids=(32, 64, 128)):
        """(ipc, process-group) microseconds per all-reduce of the gradient arena, max over ranks.  The IPC kernels'
        grid is tuned on the way (measure, don't guess: how many workgroups saturate seven xGMI links is not something
        a one-GPU build box can tell); every rank sees the same max-over-ranks timings, so all pick the same grid."""
        import time
        e = self.engine
        buf = torch.zeros_like(e.G)
        fns = [(lambda g=g: self.ipc.all_reduce(buf, blocks=g)) for g in grids] + [lambda: self._reduce(buf)]
        out = []
        for fn in fns:
            with self._ctx():
                fn()
                e.stream.synchronize()
                dist.barrier(group=self.group)
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                e.stream.synchronize()
            out.append((time.perf_counter() - t0) / iters * 1e6)
        t = torch.tensor(out, dtype=torch.float64, device="cpu" if self._stage_host else e.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        t = t.cpu().tolist()
        best = min(range(len(grids)), key=lambda i: t[i])
        self.ipc.blocks = grids[best]
        return float(t[best]), float(t[-1])

    def _try_ipc(self, required: bool) -> None:
        import socket
        where = [None] * self.world
        dev = self.engine.device
        dist.all_gather_object(where, (socket.gethostname(), dev.index if dev.index is not None else torch.cuda.current_device()),
                               group=self.group)
        ok = len(set(h for h, _d in where)) == 1 and self.world <= 8
        # ranks sharing one GPU (functional tests): a forked stream whose kernels spin on a peer that needs the same
        # CUs only slows both down (measured 0.51 vs 0.27 ms/step for 2 ranks on one GPU), so no overlap there
        self._shared_gpu = len(set(where)) < self.world
        shared_gpu = self._shared_gpu and not os.environ.get("DTF_IPC_FORCE_OVERLAP")
        err = "ranks span several hosts or more than 8 ranks"
        if ok:
            # (IpcAllReduce and self_test keep every rank on the same sequence of collectives even when a local HIP
            # call fails, and self_test's verdict is already agreed across ranks)
            from .ipc_comm import IpcAllReduce
            self.ipc = IpcAllReduce(self.engine.G.numel(), self.engine.device, group=self.group,
                                    skip_ptr=self.engine.skip_t.data_ptr() if hasattr(self.engine, "skip_t") else 0)
            ok = self.ipc.self_test()
            err = self.ipc.failure or "self-test against the process group's all-reduce failed on some rank"
            if ok:
                # measure, don't guess: tune the IPC grid on the arena-sized all-reduce.  Whether gradients then travel
                # by IPC or by the process group is decided by _tune_schedule on whole steps (the gather schedule does
                # not all-reduce the arena at all, so a slower IPC all-reduce says nothing about it).
                t_ipc, t_pg = self._time_transports(grids=(32, 64, 128) if len(set(where)) < self.world
                                                    else (64, 128, 256, 512))
                self.transport_us = {"ipc": round(t_ipc, 1), "ipc_workgroups": self.ipc.blocks,
                                     "process_group": round(t_pg, 1)}
        if ok:
            self.ipc.set_timeout(float(os.environ.get("DTF_IPC_TIMEOUT_S", "30")))
            self.engine.set_grad_hook(_IpcGradHook(self.ipc, self._bucket_a, self._bucket_b, self.engine.G,
                                                   self.overlap and not shared_gpu))
            self.comm = "ipc"
        else:
            if self.ipc is not None:
                self.ipc.close()  # (collective: the verdict is the same on every rank)
            self.ipc = None
            if required:
                raise RuntimeError("IPC all-reduce unavailable: " + err)
            if self.rank == 0:
                print("IPC all-reduce not used (%s); gradients go through %s" % (err, dist.get_backend(self.group)), flush=True)

    def _tune_schedule(self, allow_rccl: bool, steps: int = 64) -> None:
        """Pick the step schedule by running it.  Whether forking the fc-bucket all-reduce onto a second queue beats
        reducing the whole arena in-stream depends on what a cross-queue dependency costs against what the xGMI
        transfer takes at this world size (5-10 us per edge was measured on one GPU; the transfer time cannot be
        measured on a one-GPU build box), and RCCL's own overlapped path is a third candidate.  Each candidate runs
        ``steps`` real training steps from the same snapshot of (P, M, V, step), which is restored afterwards, so
        training starts from exactly the state it would have had without tuning.  All ranks see the same
        max-over-ranks timings and therefore make the same choice."""
        import time
        from . import schedule_select as sel
        e = self.engine
        forced = os.environ.get("DTF_DP_SCHEDULE", "auto")
        room = e.spe - (e.global_step % e.spe) - 1 if getattr(e, "dataset", None) is not None else steps + 8
        steps = min(steps, room - 8)
        fc1_end = e.bucket_bounds()[0][0] + e.fc1_grad_numel()  # the fc1 filter gradient leads the arena
        zc_ok = e.bucket_bounds()[0][0] == 0 and fc1_end % 4 == 0 and (e.G.numel() - fc1_end) % 4 == 0
        shared = self._shared_gpu and not os.environ.get("DTF_IPC_FORCE_OVERLAP")
        skip_ptr = e.skip_t.data_ptr() if hasattr(e, "skip_t") else 0
        op_timeout = float(os.environ.get("DTF_IPC_TIMEOUT_S", "30"))

        def exchange_buffer(n_floats: int):
            """A further IPC exchange buffer (collective set-up: every rank decides it from the same facts)."""
            from .ipc_comm import IpcAllReduce
            c = IpcAllReduce(n_floats, e.device, group=self.group, skip_ptr=skip_ptr)
            # (mapping and protocol are what can fail here; the arena-sized reduction was verified on the first buffer)
            if c.self_test(n=1 << 18, rounds=2):
                c.set_timeout(op_timeout)
                return c
            c.close()
            return None

        # "gather": exchange the fc1 filter gradient's OPERANDS (0.8 MB/rank) instead of the gradient (12.8 MB) through a
        # second, small exchange buffer
        gather_ok = zc_ok and hasattr(e, "gather_supported") and e.gather_supported() and \
            forced in ("auto", "ipc-gather", "ipc-shard", "ipc-wshard", "ipc-wshard2")
        if gather_ok and self.ipc_act is None:
            self.ipc_act = exchange_buffer(sum(e.gather_bytes()) // 4)
            gather_ok = self.ipc_act is not None
        # "shard": the same operand exchange, but every rank computes only its 1/world row slice of the global gradient and
        # pushes it into every rank's landing buffer: per-rank work independent of the world size (csrc/hip/shard_body.h)
        shard_ok = gather_ok and hasattr(e, "shard_supported") and e.shard_supported() and \
            forced in ("auto", "ipc-shard", "ipc-wshard", "ipc-wshard2")
        if shard_ok and self.ipc_w is None:
            self.ipc_w = exchange_buffer(e.fc1_grad_numel())
            shard_ok = self.ipc_w is not None
        # "wshard": the slices that travel are updated weights, the optimizer state of W_fc1 is sharded (engine.hip)
        wshard_ok = shard_ok and hasattr(e, "set_wshard") and os.environ.get("DTF_WSHARD", "1") != "0"
        # ("ipc-wshard2": a share of each rank's rows is updated and sent from the data-gradient launch, the rest from the
        # filter-gradient launch -- two launches for the stores to drain; which of the two is faster depends on the links)
        cands = sel.candidates(wshard_ok=wshard_ok and forced in ("auto", "ipc-wshard", "ipc-wshard2"),
                               shard_ok=shard_ok and forced in ("auto", "ipc-shard"),
                               gather_ok=gather_ok and forced in ("auto", "ipc-gather"), zc_ok=zc_ok, shared_gpu=shared,
                               allow_rccl=allow_rccl, host_staged=self._stage_host, forced=forced)
        self._ipc_hook, self._fc1_end = e.grad_hook, fc1_end
        use = self._use_schedule

        def set_timeouts(sec: float) -> None:  # (by-value kernel argument: takes effect at the next graph capture)
            for c in (self.ipc, self.ipc_act, self.ipc_w):
                if c is not None:
                    c.set_timeout(sec)

        trial = sel.trial_list(cands)
        timed = steps >= 8 and len(cands) > 1
        if len(trial) > 1 and room >= 2:
            cands = trial
            with self._ctx():
                snap = [t.clone() for t in (e.P, e.M, e.V, e.step_t)]
            done0 = e._steps_done
            b1 = float(e.cfg.beta1) if hasattr(e, "cfg") else float(getattr(e, "b1", 0.9))
            budget = float(os.environ.get("DTF_DP_STARTUP_BUDGET_S", "5"))
            trainer = self
            dev = "cpu" if self._stage_host else e.device

            class Ops(sel.TrialOps):
                errs_seen = 0.0

                def use(self, name):
                    use(name)

                def capture_ok(self):
                    # (also the barrier: graph capture takes a different time on every rank; the waits are short here)
                    return trainer._all_ranks_ok(trainer._capture_ok())

                def first_step(self):
                    # one step from the snapshot: Adam's first moment is linear in the gradient, m' = b1*m + (1-b1)*g, which
                    # gives the reduced gradient this schedule produced, wherever it keeps it
                    trainer.train_steps(1)
                    trainer.materialize()  # (wshard: every rank's slice of the first moment)
                    with trainer._ctx():
                        return (e.M - b1 * snap[1]) / (1.0 - b1)

                def wait_expired(self):
                    t = torch.tensor([float(trainer.transport_errors())], dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=trainer.group)
                    bad = float(t.item()) > self.errs_seen
                    self.errs_seen = max(self.errs_seen, float(t.item()))
                    return bad

                def time_steps(self):
                    trainer.train_steps(7)
                    e.stream.synchronize()
                    dist.barrier(group=trainer.group)
                    t0 = time.perf_counter()
                    trainer.train_steps(steps)
                    e.stream.synchronize()
                    return (time.perf_counter() - t0) / steps * 1e6

                def restore(self):
                    with trainer._ctx():
                        for t, s0 in zip((e.P, e.M, e.V, e.step_t), snap):
                            t.copy_(s0)
                        e.metrics_ring.zero_()  # (the per-step loss/accuracy slots the trial steps accumulated into)
                        if hasattr(e, "skip_t"):
                            e.skip_t.zero_()
                    e.stream.synchronize()
                    e._steps_done = done0
                    trainer._wsh_reload()

                def over_budget(self):
                    """Start-up (self-tests + trials so far) has used its budget on some rank (agreed across ranks)."""
                    t = torch.tensor([1.0 if time.perf_counter() - trainer._t_start > budget else 0.0], dtype=torch.float64,
                                     device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=trainer.group)
                    return float(t.item()) > 0

                def deviation(self, g, ref):
                    with trainer._ctx():
                        return float((g - ref).norm() / (ref.norm() + 1e-30))

                def reduce_max(self, values):
                    t = torch.tensor(values, dtype=torch.float64, device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=trainer.group)
                    return t.cpu().tolist()

            set_timeouts(2.0)  # a schedule that cannot make progress on this machine must cost seconds, not minutes
            times, off, broken = sel.run_trials(cands, Ops(), forced=forced, timed=timed, note=self._note, budget_s=budget)
            self.schedule_us = {c: round(v, 1) for c, v in zip(cands, times)} if timed else None
            self.schedule_deviation = {c: float("%.3g" % v) for c, v in zip(cands, off)}
            set_timeouts(op_timeout)
            cands = [sel.choose(cands, times, off, broken, forced)]
        use(cands[0])
        if self.rank == 0:
            print("data-parallel step schedule: %s %s gradient deviation from the staged all-reduce %s" % (
                self.schedule, self.schedule_us or "", getattr(self, "schedule_deviation", "")), flush=True)

    def _use_schedule(self, c: str) -> None:
        e, hook, fc1_end = self.engine, self._ipc_hook, self._fc1_end
        self._wsh_leave()  # (collective, like every change of schedule)
        zc, wsh = c == "ipc-zerocopy", c in ("ipc-wshard", "ipc-wshard2")
        sh = c == "ipc-shard" or wsh
        ga = c == "ipc-gather" or sh  # (the shard schedules ride on the gather schedule's operand exchange and publish)
        if self.ipc is not None or zc:
            e.set_grad_redirect(self.ipc.stage_ptr if zc else 0, self.ipc.result_ptr if zc else 0)
        if hasattr(e, "set_shard"):
            e.set_shard(None)  # (also switches the gather exchange off)
        if sh:
            if wsh:
                e.set_wshard(self.ipc_act, self.ipc_w,
                             split=float(os.environ.get("DTF_WSHARD_SPLIT", "0.45")) if c == "ipc-wshard2" else 0.0)
            else:
                e.set_shard(self.ipc_act, self.ipc_w)
            with torch.cuda.stream(e.stream):
                e.G[:fc1_end].zero_()  # this rank's slice of the fc1 gradient is accumulated with atomics and re-zeroed by its reader
        elif hasattr(e, "set_gather"):
            e.set_gather(self.ipc_act if ga else None)
        self._rccl_graph = False
        if c == "rccl":
            if self.rccl is not None:  # native communicator: the collectives are part of the step graph
                (a0, a1), (b0, b1) = e.bucket_bounds()
                e.set_grad_hook(_RcclGradHook(self.rccl, e.G[a0:a1], e.G[b0:b1], e.G,
                                              self.overlap and os.environ.get("DTF_RCCL_OVERLAP", "1") != "0"))
                self._rccl_graph = True
            else:
                e.set_grad_hook(None)
            self.comm = "rccl"
        else:
            hook.set_overlap(c == "ipc-overlap")
            hook.set_zero_copy(fc1_end if zc else 0)
            hook.set_gather(fc1_end if ga else 0)
            e.set_grad_hook(hook)  # (drops the captured graphs)
            self.comm = "ipc"
        self.schedule = c
        if wsh:
            self._wsh_enter()

    # ------------------------------------------------------------------ "ipc-wshard": sharded state
    def _wsh_enter(self) -> None:
        """The arenas are consistent and identical on every rank here (start-up, or right after a restore)."""
        e = self.engine
        e.wshard_load()
        self._wsh = True
        self._wsh_dirty = False  # (the arenas ARE the state until the next step runs)
        self._wsh_snapshot()
        e.state_provider = self._wsh_state
        e.on_state_loaded = self._wsh_reload

    def _wsh_reload(self) -> None:
        """The arenas were rewritten from outside (checkpoint restore, broadcast, trial snapshot): W_fc1's buffers follow."""
        if self._wsh:
            self.engine.wshard_load()
            self._wsh_dirty = False
            self._wsh_snapshot()

    def _wsh_snapshot(self) -> None:
        e = self.engine
        with e._capture_lock:
            with torch.cuda.stream(e.stream):
                snap = {"P": e.P.clone(), "M": e.M.clone(), "V": e.V.clone(), "step": e.step_t.clone()}
            e.stream.synchronize()
        self._wsh_snap = snap

    def _wsh_state(self):
        """What ``engine.state_dict()`` returns under this schedule: the last state every rank agreed on (at most
        ``DTF_WSHARD_SNAPSHOT_STEPS`` steps old), not the live arenas, whose optimizer rows of the other ranks are stale."""
        return {k: v.clone() for k, v in self._wsh_snap.items()}

    def sync_params(self) -> None:
        """Make ``engine.P`` current (local, stream-ordered; a no-op for every schedule that keeps it current anyway)."""
        if self._wsh:
            self.engine.wshard_store()

    def materialize(self, snapshot: bool = False) -> None:
        """Collective.  Under "ipc-wshard": bring the live arenas to the consistent state of this step — W_fc1 from its
        current buffer, every rank's rows of M and V from their owner (process-group broadcasts: this is not on the step's
        path) — and optionally keep it as the checkpoint snapshot.  A no-op for every other schedule."""
        if not self._wsh:
            return
        if not self._wsh_dirty:  # (no step since the arenas were last made whole: same verdict on every rank)
            if snapshot:
                self._wsh_snapshot()
            return
        e = self.engine
        e.wshard_store()
        n = e.fc1_grad_numel()
        cols = n // 3136
        with self._ctx():
            for r in range(self.world):
                lo, hi = e.shard_rows(r, self.world)
                for t in (e.M, e.V):
                    self._bcast(t[lo * cols:hi * cols], r)
        self._wsh_dirty = False
        if snapshot:
            self._wsh_snapshot()

    def _wsh_leave(self) -> None:
        if not self._wsh:
            return
        self.materialize()
        e = self.engine
        e.stream.synchronize()
        self._wsh = False
        self._wsh_snap = None
        e.state_provider = None
        e.on_state_loaded = None

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
