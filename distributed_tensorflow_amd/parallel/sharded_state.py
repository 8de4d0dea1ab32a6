"""State handling of the "ipc-wshard" data-parallel schedules (``csrc/hip/engine.hip``: ``MnistEngine::wshard``), where W_fc1
lives in IPC buffers and its optimizer state is sharded over the ranks: the live arenas of a rank are then not a whole
state.  This mixin of ``DataParallelTrainer`` assembles it (``materialize``: collective), keeps the last state every rank
agreed on for checkpoints (``engine.state_dict()`` is redirected to it) and follows rewrites of the arenas from outside
(restore, broadcast, trial snapshots).  Reference behaviour being replaced: the variables live whole on the parameter
server and ``tf.train.Saver`` reads them there (``/root/reference/demo2/train.py:27-29,166-172``)."""
from __future__ import annotations

import contextlib

import torch


class ShardedStateMixin:
    def _wsh_on_stream(self):
        """Context in which the engine's tensors are read: its stream, under its capture lock (a CPU stand-in has neither)."""
        e = self.engine
        stack = contextlib.ExitStack()
        stack.enter_context(getattr(e, "_capture_lock", contextlib.nullcontext()))
        if hasattr(e, "stream"):
            stack.enter_context(torch.cuda.stream(e.stream))
            stack.callback(e.stream.synchronize)  # (runs first on exit: the copies have landed before the lock is released)
        return stack

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
        with self._wsh_on_stream():
            snap = {"P": e.P.clone(), "M": e.M.clone(), "V": e.V.clone(), "step": e.step_t.clone()}
        self._wsh_snap = snap

    def _wsh_state(self):
        """What ``engine.state_dict()`` returns under this schedule: the last state every rank agreed on (at most
        ``DTF_WSHARD_SNAPSHOT_STEPS`` steps old), not the live arenas, whose optimizer rows of the other ranks are stale.
        Called from the Supervisor's saver thread as well: like ``engine.state_dict()`` it works on the engine's stream and
        never while the training thread has a graph capture open."""
        with self._wsh_on_stream():
            snap = self._wsh_snap  # (replaced as a whole by _wsh_snapshot, never modified in place)
            out = {k: v.clone() for k, v in snap.items()}
        return out

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
        cols = e.fc1_grad_numel() // e.shard_rows(self.world - 1, self.world)[1]  # (the last rank's slice ends at the last row)
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
        if hasattr(e, "stream"):
            e.stream.synchronize()
        self._wsh = False
        self._wsh_snap = None
        e.state_provider = None
        e.on_state_loaded = None
