"""Start-up of the data-parallel trainer: which transports work on this machine (native RCCL communicator, IPC exchange
buffers with their self-tests), how fast they are, and which step schedule runs — decided by trial steps from a snapshot
(``schedule_select.py`` holds the decision logic).  A mixin of ``DataParallelTrainer``; every method here is collective.
Reference behaviour being replaced: a fixed star of gRPC channels to the parameter server
(``/root/reference/demo2/train.py:13-29``); there is nothing to choose."""
from __future__ import annotations

import math
import os

import torch
import torch.distributed as dist

from .grad_hooks import IpcGradHook, RcclGradHook


class StartupMixin:
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
            self.engine.set_grad_hook(IpcGradHook(self.ipc, self._bucket_a, self._bucket_b, self.engine.G,
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
            forced in ("auto", "ipc-gather", "ipc-gather2", "ipc-shard", "ipc-wshard", "ipc-wshard2")
        # "ipc-gather2" (fp32 engine): pool2 is PUSHED to the peers during the fc1 forward launch, only dh is exchanged after
        # the loss head; the exchange buffer then also holds one landing block per rank
        gather2_ok = gather_ok and getattr(e, "dtype", "") == "fp32" and os.environ.get("DTF_EARLY_PUSH", "1") != "0" and \
            forced in ("auto", "ipc-gather2")
        if gather_ok and self.ipc_act is None:
            n_act = sum(e.gather_bytes()) // 4
            if gather2_ok:
                n_act = max(n_act, self.world * (e.gather_bytes()[0] // 4))
            self.ipc_act = exchange_buffer(n_act)
            gather_ok = self.ipc_act is not None
        gather2_ok = gather2_ok and gather_ok
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
                               gather_ok=gather_ok and forced in ("auto", "ipc-gather"), gather2_ok=gather2_ok,
                               zc_ok=zc_ok, shared_gpu=shared,
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
            budget = float(os.environ.get("DTF_DP_STARTUP_BUDGET_S", "10"))
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
                    # (fault injection for tests/test_rccl_gpu.py: the reference schedule's waits "expire", every time)
                    if os.environ.get("DTF_DP_FAULT_INJECT") == "reference-expires" and trainer.schedule == sel.REFERENCE:
                        bad = True
                    return bad

                def time_steps(self):
                    """Warm + timed steps -> us per step; inf (agreed across ranks) as soon as an in-kernel wait has
                    expired: a schedule whose waits run into their 2 s bound would otherwise hold the start-up for
                    minutes (seen with four ranks time-sharing one GPU: 64 steps took 85 s)."""
                    for n in (2, 5):
                        trainer.train_steps(n)
                        e.stream.synchronize()
                        if self.wait_expired():
                            return math.inf
                    dist.barrier(group=trainer.group)
                    done, spent = 0, 0.0
                    while done < steps:
                        n = min(16, steps - done)
                        t0 = time.perf_counter()
                        trainer.train_steps(n)
                        e.stream.synchronize()
                        spent += time.perf_counter() - t0
                        done += n
                        if self.wait_expired():  # (also the barrier in front of the next chunk; not part of the timing)
                            return math.inf
                    return spent / steps * 1e6

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
            ops = Ops()
            times, off, broken = sel.run_trials(cands, ops, forced=forced, timed=timed, note=self._note, budget_s=budget)
            self.schedule_us = {c: round(v, 1) for c, v in zip(cands, times)} if timed else None
            self.schedule_deviation = {c: float("%.3g" % v) for c, v in zip(cands, off)}
            set_timeouts(op_timeout)
            try:
                cands = [sel.choose(cands, times, off, broken, forced)]
            except RuntimeError as err:  # (with what the trials saw on this rank)
                # (what the trials saw on this rank: norms of the first-step gradients, the exchanges' device words)
                seen = ["%s |g| %.4g" % (c, float(g.norm())) for c, g in zip(cands, getattr(ops, "grads", [])) if g is not None]
                words = [c._words.tolist() for c in (self.ipc, self.ipc_act, self.ipc_w) if c is not None]
                raise RuntimeError("%s; rank %d: first-step gradients %s; [epoch, errors, ticket, seen, tag, ..] of the "
                                   "exchanges %s; start-up log: %s" % (err, self.rank, ", ".join(seen), words,
                                                                       " | ".join(self.startup_log))) from None
            # the dropped candidates' expired waits are in the log and the state is the snapshot again: the chosen schedule
            # starts with a clean count (ensure_healthy() / check_transport() act on ANY expiry since then)
            if broken:
                for comm in (self.ipc, self.ipc_act, self.ipc_w):
                    if comm is not None:
                        comm.reset_errors()
                dist.barrier(group=self.group)
        use(cands[0])
        if self.rank == 0:
            print("data-parallel step schedule: %s %s gradient deviation from the staged all-reduce %s" % (
                self.schedule, self.schedule_us or "", getattr(self, "schedule_deviation", "")), flush=True)

    def _use_schedule(self, c: str) -> None:
        e, hook, fc1_end = self.engine, self._ipc_hook, self._fc1_end
        self._wsh_leave()  # (collective, like every change of schedule)
        zc, wsh = c == "ipc-zerocopy", c in ("ipc-wshard", "ipc-wshard2")
        sh = c == "ipc-shard" or wsh
        ga = c in ("ipc-gather", "ipc-gather2") or sh  # (the shard schedules ride on the gather schedule's exchange and publish)
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
            if c == "ipc-gather2":
                e.set_gather(self.ipc_act, early_push=True)
            else:
                e.set_gather(self.ipc_act if ga else None)
        self._rccl_graph = False
        if c == "rccl":
            if self.rccl is not None:  # native communicator: the collectives are part of the step graph
                (a0, a1), (b0, b1) = e.bucket_bounds()
                e.set_grad_hook(RcclGradHook(self.rccl, e.G[a0:a1], e.G[b0:b1], e.G,
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
