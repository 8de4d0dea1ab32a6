"""Supervisor (component C23): chief election, init/restore, wait-for-ready, timed autosave,
recovery and the stop flag.

Reference: ``tf.train.Supervisor(is_chief=(task_index==0), logdir="./logs", init_op, saver,
global_step, save_model_secs=600)`` + ``sv.managed_session(server.target)`` +
``sv.should_stop()`` / ``sv.stop()`` (``demo2/train.py:166-176,181``;
``retrain2/retrain2.py:423-431,445,508``).

Equivalent behaviour here:

* the chief (rank 0) either restores the newest checkpoint in ``logdir`` (auto-resume after a
  crash/restart) or keeps its fresh initialisation; the result is **broadcast** to every replica —
  the replacement for "chief runs init_op, other workers poll until the variables are initialised";
* a background thread on the chief snapshots the model every ``save_model_secs`` seconds to
  ``logdir/model.ckpt-<global_step>`` (device->host copy on a side stream, file I/O off the
  training thread) and, like the Supervisor's step-counter thread, logs ``global_step/sec``;
* ``managed_session`` is a context manager that turns any exception into a coordinated stop
  (``request_stop``) and always runs a final save on the chief (the reference's demo2 never saves
  explicitly, quirk Q8);
* the stop flag and liveness travel OUT OF BAND, through the job's rendezvous store
  (``csrc/host/tcp_store.cc``): ``request_stop`` sets a store key, every rank's monitor thread polls
  it and bumps its heartbeat counter (``cluster.Heartbeat``), and a rank whose counter stops moving
  for ``2 x period`` turns into a stop request on every survivor.  ``should_stop()`` then only reads
  a local flag — no collective, so a rank that died cannot wedge it.  A survivor that is itself
  blocked inside a collective on the dead rank (RCCL has no per-call timeout) cannot poll anything:
  if the training thread has not DRAINED (stream synchronised, final checkpoint written) within
  ``fatal_grace_s`` of the stop the monitor ends the process with exit status 70, and the relaunched
  job resumes from the last checkpoint.  Seeing the flag is not enough to disarm that watchdog:
  a survivor that has already enqueued collectives its peer never joins blocks in its NEXT stream
  sync (summary flush, final save), after ``should_stop()`` has returned True;
* a stop that nobody died for (``request_stop()`` without an exception, a signal, the step budget
  of a peer) is AGREED: every rank publishes ``its step + stop_margin_steps`` under
  ``sv/vote/<rank>`` when it first sees the flag and keeps training to the maximum of the votes, so
  all ranks enqueue the same number of collectives and every one of them drains.  A rank that
  failed publishes a negative vote instead, and the survivors leave at once (watchdog armed);
* without a store (single process, or a launcher that has none) the local flag is all-reduced (max)
  every ``stop_check_interval`` STEPS;
* ``--fault_inject_step`` support lives in the training drivers and is exercised by the tests.
"""
from __future__ import annotations

import contextlib
import os
import threading
import time
from typing import Callable, Optional

from ..utils import checkpoint as ckpt


class ConsistencyError(RuntimeError):
    """A check of the replicas' state failed on this rank (checksums diverged, non-finite loss ...)."""


def state_is_suspect(ex: Optional[BaseException]) -> bool:
    """Whether a rank that left its loop on ``ex`` may still checkpoint.  A peer's failure, a stop request and a transport
    error leave the parameters at the last completed step (they only change after a completed exchange); a failed
    consistency check, an assertion of the exchange check, a native kernel error or a floating-point error do not."""
    if ex is None:
        return False
    from ..native import NativeError
    return isinstance(ex, (ConsistencyError, AssertionError, NativeError, FloatingPointError, ArithmeticError))


class Supervisor:
    def __init__(self, trainer, *, is_chief: bool, logdir: Optional[str], saver: Optional[ckpt.Saver] = None,
                 save_model_secs: float = 600.0, checkpoint_basename: str = "model.ckpt",
                 stop_check_interval: int = 50, summary_writer=None, log_fn: Callable[[str], None] = print,
                 store=None, heartbeat_period_s: float = 1.0, fatal_grace_s: float = 20.0,
                 stop_margin_steps: int = 64):
        self.trainer = trainer
        self.engine = trainer.engine
        self.is_chief = is_chief
        self.logdir = logdir
        self.saver = saver or ckpt.Saver(self.engine, with_global_step=True)
        self.save_model_secs = save_model_secs
        self.save_path = os.path.join(logdir, checkpoint_basename) if logdir else None
        self.stop_check_interval = max(1, stop_check_interval)
        self.summary_writer = summary_writer
        self.log = log_fn
        self._stop_local = threading.Event()
        self._stop_global = False
        self._steps_since_check = 0
        self._saver_thread: Optional[threading.Thread] = None
        self._saver_wake = threading.Event()
        self._save_lock = threading.Lock()
        self.restored_from: Optional[str] = None
        self.num_autosaves = 0
        self.last_exception: Optional[BaseException] = None
        # out-of-band control plane (None: collective stop flag only)
        self.store = store
        self.heartbeat_period_s = heartbeat_period_s
        self.fatal_grace_s = fatal_grace_s
        self.heartbeat = None
        self.dead_ranks: list = []
        self.stop_reason: Optional[str] = None
        self._monitor_thread: Optional[threading.Thread] = None
        self._failure_seen_at: Optional[float] = None
        self._acknowledged = False  # True once this rank has drained (nothing left that could block on a peer)
        self._stop_key = "sv/stop"
        self.peer_failed = False  # a peer raised, died or never voted: no further collective with it
        # agreed stop step (graceful stops only)
        self.stop_margin_steps = max(1, stop_margin_steps)
        self._vote: Optional[int] = None
        self.stop_at: Optional[int] = None

    # ------------------------------------------------------------------ init / restore
    def prepare_or_wait_for_session(self) -> int:
        """Chief: restore the newest checkpoint if any; everyone: receive the chief's state."""
        if self.is_chief and self.logdir:
            os.makedirs(self.logdir, exist_ok=True)
            latest = ckpt.latest_checkpoint(self.logdir, os.path.basename(self.save_path))
            if latest is not None:
                step = self.saver.restore(latest)
                self.restored_from = latest
                self.log("Supervisor: restored %s (global_step=%d)" % (latest, step))
        if self.trainer.world > 1:
            self.trainer.broadcast_state(src=0)
        return self.engine.global_step

    # ------------------------------------------------------------------ autosave
    def _save(self, tag: str) -> Optional[str]:
        if not (self.is_chief and self.save_path):
            return None
        with self._save_lock:
            path = self.saver.save(self.save_path, global_step=self.engine.global_step)
        self.log("Supervisor: %s checkpoint %s" % (tag, path))
        return path

    def _saver_loop(self) -> None:
        last_step, last_t = self.engine.global_step, time.time()
        while not self._stop_local.is_set():
            woke = self._saver_wake.wait(self.save_model_secs)
            if self._stop_local.is_set() or woke:
                break
            try:
                self._save("autosave")
                self.num_autosaves += 1
                now, step = time.time(), self.engine.global_step
                rate = (step - last_step) / max(1e-9, now - last_t)
                last_step, last_t = step, now
                if self.summary_writer is not None:
                    self.summary_writer.add_scalars({"global_step/sec": rate}, step)
            except Exception as e:  # never kill training because of a failed autosave
                self.log("Supervisor: autosave failed: %r" % (e,))

    def start_standard_services(self) -> None:
        if self.is_chief and self.save_path and self.save_model_secs and self.save_model_secs > 0:
            self._saver_thread = threading.Thread(target=self._saver_loop, name="dtf-saver", daemon=True)
            self._saver_thread.start()
        if self.store is not None and self.trainer.world > 1 and self._monitor_thread is None:
            from .cluster import Heartbeat
            rank = getattr(self.trainer, "rank", 0)
            self.heartbeat = Heartbeat(self.store, rank, period_s=self.heartbeat_period_s).start()
            self._monitor_thread = threading.Thread(target=self._monitor_loop, name="dtf-monitor", daemon=True)
            self._monitor_thread.start()

    # ------------------------------------------------------------------ failure detection (out of band)
    def _monitor_loop(self) -> None:
        """Poll the store: the job-wide stop key and every rank's heartbeat.  Runs on every rank."""
        world = self.trainer.world
        period = self.heartbeat_period_s
        t_start = time.time()
        while not self._saver_wake.is_set():
            try:
                stop = self.store.counter(self._stop_key) > 0
                # (a rank is only declared dead after it has been seen alive: start-up skew is not a failure)
                dead = self.heartbeat.dead_ranks(world, 2.0 * period) if time.time() - t_start > 2.0 * period else []
                dead = [r for r in dead if self.store.counter("hb/%d" % r) > 0 and self.store.counter("sv/done/%d" % r) == 0]
            except ConnectionError:
                stop, dead = True, []
                self.stop_reason = self.stop_reason or "rendezvous store unreachable"
            if dead and not self.dead_ranks:
                self.dead_ranks = dead
                self.stop_reason = "rank(s) %s stopped sending heartbeats" % dead
                self.log("Supervisor: " + self.stop_reason)
                try:
                    self.store.add(self._stop_key, 1)
                except ConnectionError:
                    pass
                stop = True
            if stop and not self._stop_local.is_set():
                self.stop_reason = self.stop_reason or "stop requested by a peer"
                self._stop_local.set()
            if stop and self._failure_seen_at is None:
                self._failure_seen_at = time.time()
            if (self._failure_seen_at is not None and not self._acknowledged and self.fatal_grace_s > 0
                    and time.time() - self._failure_seen_at > self.fatal_grace_s):
                # the training thread never came back to look at the flag: it is blocked in a collective on a dead peer
                self.log("Supervisor: %s; training thread unresponsive for %.0f s -> exiting (status 70), "
                         "relaunch resumes from the last checkpoint" % (self.stop_reason, self.fatal_grace_s))
                import sys
                for f in (sys.stdout, sys.stderr):  # (os._exit skips the interpreter's own flush)
                    try:
                        f.flush()
                    except Exception:
                        pass
                os._exit(70)
            time.sleep(period / 2.0)

    # ------------------------------------------------------------------ stop protocol
    def request_stop(self, ex: Optional[BaseException] = None) -> None:
        if ex is not None:
            self.last_exception = ex
            self.stop_reason = self.stop_reason or "exception on this rank: %r" % (ex,)
        self._stop_local.set()
        if self.store is not None:
            try:
                if ex is not None and self.trainer.world > 1:
                    # this rank leaves its loop NOW: the peers must not wait for it at an agreed step
                    self._vote = -1
                    self.store.set("sv/vote/%d" % getattr(self.trainer, "rank", 0), b"-1")
                self.store.add(self._stop_key, 1)  # out of band: the peers learn of it without a collective
            except ConnectionError:
                pass

    def _read_votes(self):
        """Votes of all ranks, or None while one is missing.  A negative vote = that rank failed."""
        votes = []
        for r in range(self.trainer.world):
            v = self.store.get("sv/vote/%d" % r, timeout_s=0.0)
            if v is None:
                return None
            votes.append(int(v.decode()))
        return votes

    def _agree_stop_step(self, step: int) -> bool:
        """The flag is up and nobody is known to have failed: vote, then train on to the maximum of the votes.  Returns
        True when this rank has to leave its loop now."""
        rank = getattr(self.trainer, "rank", 0)
        if self._vote is None:
            self._vote = step + self.stop_margin_steps
            self.store.set("sv/vote/%d" % rank, str(self._vote).encode())
        t0 = time.time()
        while self.stop_at is None:
            votes = None if self.dead_ranks else self._read_votes()
            if self.dead_ranks or (votes is not None and min(votes) < 0):
                self.peer_failed = True
                return True  # someone is gone: no agreement possible, leave (the watchdog stays armed until drained)
            if votes is not None:
                self.stop_at = max(votes)
            elif step < self._vote:
                return False  # keep training below the own vote; look again at the next call
            elif time.time() - t0 > max(1.0, self.fatal_grace_s):
                self.peer_failed = True
                return True  # a peer never voted (it is blocked or gone): same as a failure
            else:
                time.sleep(min(0.05, self.heartbeat_period_s / 4.0))
        return step >= self.stop_at

    def steps_allowed(self, step: int, want: int) -> int:
        """How many of ``want`` steps a driver that launches several steps at once may still enqueue (an agreed stop
        step must not be overshot; while the votes are still coming in the own vote is the limit)."""
        limit = self.stop_at if self.stop_at is not None else self._vote
        if limit is None or limit < 0:
            return want
        return max(1, min(want, limit - step))

    def should_stop(self, force_check: bool = False, steps: int = 1) -> bool:
        """With a rendezvous store: reads the local flag the monitor thread maintains and, for a stop nobody failed
        for, the peers' votes from the store (no collective either way).  Without: collective — every rank calls it at
        the same points — and the flag is all-reduced every ``stop_check_interval`` STEPS (``steps`` = steps run since
        the previous call).  Returning True does NOT disarm the watchdog: only a completed drain (``stop()``) does."""
        if self._stop_global:
            return True
        if self.store is not None and self.trainer.world > 1:
            if self._stop_local.is_set():
                try:
                    failed = self.last_exception is not None or bool(self.dead_ranks) or (self._vote is not None and self._vote < 0)
                    self._stop_global = True if failed else self._agree_stop_step(self.engine.global_step)
                    self.peer_failed = self.peer_failed or failed
                except ConnectionError:
                    self._stop_global = True
                    self.peer_failed = True
            return self._stop_global
        self._steps_since_check += max(1, steps)
        if force_check or self._steps_since_check >= self.stop_check_interval or self.trainer.world == 1:
            self._steps_since_check = 0
            self._stop_global = self.trainer.any_rank(self._stop_local.is_set())
        return self._stop_global

    def clean_stop(self) -> bool:
        """Every rank left (or will leave) its loop at the same step and nobody failed: training ran to its end, or a stop
        was agreed.  Only then may the drivers run one more collective (e.g. assembling sharded optimizer state for the
        final checkpoint) before ``stop()``."""
        return self.last_exception is None and not self.dead_ranks and not self.peer_failed and \
            not (self._vote is not None and self._vote < 0)

    def _drain(self) -> None:
        """Wait for everything this rank has enqueued.  Unbounded by itself (RCCL has no per-call timeout): while it
        blocks the watchdog is still armed and ends the process with status 70 after ``fatal_grace_s``."""
        sync = getattr(self.trainer, "synchronize", None)
        if sync is not None:
            sync()
        elif hasattr(self.engine, "stream"):
            self.engine.stream.synchronize()

    def stop(self, final_save: bool = True) -> None:
        """``sv.stop()``: drain, final save on the chief, then stop the services.  The monitor (and with it the exit-70
        watchdog) lives until the save is done: both the drain and ``state_dict()`` synchronise the stream, and a peer
        that left its loop earlier never joins the collectives enqueued there."""
        self._stop_local.set()
        self._drain()
        if final_save and self.is_chief and state_is_suspect(self.last_exception):
            # the relaunch must not restore a state that is known to be wrong: it resumes from the last autosave instead
            self.log("Supervisor: no final checkpoint, this rank stopped on %r and its state is suspect" % (self.last_exception,))
        elif final_save:
            try:
                self._save("final")
            except Exception as e:
                self.log("Supervisor: final save failed: %r" % (e,))
        self._acknowledged = True  # nothing below can block on a peer
        self._saver_wake.set()
        if self._saver_thread is not None:
            self._saver_thread.join(timeout=30)
            self._saver_thread = None
        if self._monitor_thread is not None:
            self._monitor_thread.join(timeout=5)
            self._monitor_thread = None
        if self.store is not None and self.trainer.world > 1:
            # this rank is done (its heartbeat may now stop without being a failure); the chief, which usually hosts the
            # store, stays until everyone has said so
            try:
                self.store.add("sv/done/%d" % getattr(self.trainer, "rank", 0), 1)
                self.store.add("sv/done", 1)
                if self.is_chief and not self.dead_ranks:
                    self.store.wait_counter("sv/done", self.trainer.world, timeout_s=30.0)
            except ConnectionError:
                pass
        if self.summary_writer is not None:
            self.summary_writer.flush()

    @contextlib.contextmanager
    def managed_session(self, start_services: bool = True):
        """``with sv.managed_session(server.target) as sess`` — yields the trainer."""
        self.prepare_or_wait_for_session()
        if start_services:
            self.start_standard_services()
        try:
            yield self.trainer
        except BaseException as e:  # noqa: BLE001 - mirror Supervisor: any error requests a stop
            self.request_stop(e)
            self.stop(final_save=True)
            raise
        else:
            self.stop(final_save=True)
