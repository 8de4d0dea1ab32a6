"""Decision logic of the data-parallel start-up, free of GPU calls so that it can be exercised on a CPU with injected
timings, capture failures and expired waits (``tests/test_schedule_select_cpu.py``): on the 8-GPU box a wrong branch here
is otherwise found only by the run that was supposed to be measured.

``DataParallelTrainer._tune_schedule`` supplies the operations (switch schedule, capture, run a step, time steps, restore
the snapshot, agree across ranks); everything that DECIDES lives here:

* which schedules are candidates on this machine (``candidates``);
* the trial loop: bounded start-up budget, a candidate whose graph capture failed or whose in-kernel wait expired is not
  eligible, the reference schedule is always tried (``run_trials``);
* the choice: fastest among the candidates whose reduced gradient agrees with the reference schedule's, RCCL as the last
  resort, an error when nothing is usable (``choose``).

Reference behaviour being replaced: ``/root/reference/demo2/train.py:27-29,146-149,181-184`` (variables on the parameter
server, every worker pushes 13.1 MB of gradients per step; there is nothing to select).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

REFERENCE = "ipc-inline"   # the staged in-stream all-reduce: what the start-up self-test validated against the process group
TOLERANCE = 0.02           # relative deviation of a schedule's reduced gradient from the reference schedule's

# order of preference when nothing is timed
ORDER = ("ipc-wshard", "ipc-wshard2", "ipc-shard", "ipc-gather2", "ipc-gather", "ipc-zerocopy", "ipc-overlap", "ipc-inline", "rccl")


def candidates(*, shard_ok: bool, gather_ok: bool, zc_ok: bool, shared_gpu: bool, allow_rccl: bool, host_staged: bool,
               forced: str = "auto", wshard_ok: bool = False, gather2_ok: bool = False) -> List[str]:
    """Schedules worth trying here, in order of preference.  ``forced`` (``DTF_DP_SCHEDULE``) narrows the list to one
    (``ValueError`` if that one is not available)."""
    cands = (["ipc-wshard", "ipc-wshard2"] if wshard_ok else []) + (["ipc-shard"] if shard_ok else []) + \
            (["ipc-gather2"] if gather2_ok else []) + (["ipc-gather"] if gather_ok else []) + \
            (["ipc-zerocopy"] if zc_ok else []) + ([] if shared_gpu else ["ipc-overlap"]) + [REFERENCE] + \
            (["rccl"] if allow_rccl and not host_staged and not shared_gpu else [])
    if forced != "auto":
        if forced not in cands:
            raise ValueError("DTF_DP_SCHEDULE=%s is not available here (%s)" % (forced, cands))
        cands = [forced]
    return cands


def trial_list(cands: Sequence[str]) -> List[str]:
    """A forced schedule other than the reference is still checked against it."""
    cands = list(cands)
    return cands if len(cands) > 1 or cands == [REFERENCE] else cands + [REFERENCE]


class TrialOps:
    """What the trial loop needs from the trainer.  Every method that returns a verdict returns the one AGREED across
    ranks (the trainer reduces it), so all ranks walk through the same branches."""

    def use(self, name: str) -> None: raise NotImplementedError
    def capture_ok(self) -> bool: raise NotImplementedError            # graphs of the current schedule captured on every rank
    def first_step(self): raise NotImplementedError                    # run one step from the snapshot -> its reduced gradient
    def wait_expired(self) -> bool: raise NotImplementedError          # an in-kernel wait expired on some rank since last asked
    def time_steps(self) -> float: raise NotImplementedError           # warm + timed steps -> us per step (this rank)
    def restore(self) -> None: raise NotImplementedError
    def over_budget(self) -> bool: raise NotImplementedError
    def deviation(self, grad, ref) -> float: raise NotImplementedError  # relative, this rank
    def reduce_max(self, values: List[float]) -> List[float]: raise NotImplementedError


def run_trials(cands: Sequence[str], ops: TrialOps, *, forced: str = "auto", timed: bool = True,
               note: Callable[[str], None] = lambda m: None, clock: Optional[Callable[[], float]] = None,
               budget_s: float = 5.0) -> Tuple[List[float], List[float], set]:
    """Run every candidate from the same snapshot.  Returns (us per step, gradient deviation from the reference schedule,
    names that are not eligible), the first two max-reduced over ranks and aligned with ``cands``."""
    import time as _time
    clock = clock or _time.perf_counter
    times: List[float] = []
    grads: list = []
    broken: set = set()

    def drop(c: str, why: str, restore: bool = False) -> None:
        note("%s: %s" % (c, why))
        broken.add(c)
        times.append(math.inf)
        if restore:
            ops.restore()
        else:
            grads.append(None)

    for c in cands:
        # bounded start-up: once the budget is spent only the reference schedule is still tried
        if times and c != REFERENCE and forced == "auto" and ops.over_budget():
            drop(c, "not tried (start-up budget of %.0f s spent)" % budget_s)
            continue
        t_c = clock()
        ops.use(c)
        if not ops.capture_ok():
            drop(c, "not eligible (graph capture failed on some rank)")
            continue
        grads.append(ops.first_step())
        expired = c != "rccl" and ops.wait_expired()
        if expired and c == REFERENCE:
            # nothing can be chosen without the reference's gradient, and ranks that time-share a GPU can miss the bound
            # once without anything being wrong: one more try from the snapshot
            note("%s: an in-kernel wait expired (2 s bound) in the first step; trying once more" % c)
            ops.restore()
            grads[-1] = ops.first_step()
            expired = ops.wait_expired()
        if expired:
            drop(c, "an in-kernel wait expired (2 s bound): not eligible", restore=True)
            continue
        if timed:
            t_us = ops.time_steps()
            if math.isinf(t_us):  # (agreed across ranks: TrialOps.time_steps stops at the first expired wait)
                drop(c, "an in-kernel wait expired during the timed steps (2 s bound): not eligible", restore=True)
                continue
            times.append(t_us)
        else:
            times.append(float(len(times)))  # untimed: keep the order of preference
        note("%s: %s us/step, trial took %.2f s" % (c, ("%.1f" % times[-1]) if timed else "untimed", clock() - t_c))
        ops.restore()
    # every schedule computes the same gradient up to summation order (same batch, same dropout mask: both are keyed by
    # the step): a candidate that is far from the reference schedule is broken on this machine
    ops.grads = grads  # (for the caller's error report)
    ref = grads[list(cands).index(REFERENCE)] if REFERENCE in cands else None
    if REFERENCE in broken and math.isinf(times[list(cands).index(REFERENCE)]) and any(c not in broken for c in cands):
        # The reference's own exchange ran into the bound (seen with three processes time-sharing one GPU, twice in a row),
        # so its "gradient" is that of a skipped step.  The process group's all-reduce of the same local gradients is the
        # same sum: take the reference from there rather than refuse every candidate.
        ref = None
        if "rccl" in cands and "rccl" not in broken:
            ref = grads[list(cands).index("rccl")]
        else:
            note("%s: not usable as the reference; taking the reference gradient through the process group" % REFERENCE)
            ops.use("rccl")
            if ops.capture_ok():
                ref = ops.first_step()
                ops.restore()
    off = [math.inf if (g is None or ref is None) else ops.deviation(g, ref) for g in grads]
    both = ops.reduce_max(list(times) + off)
    return both[:len(times)], both[len(times):], broken


def choose(cands: Sequence[str], times: Sequence[float], off: Sequence[float], broken: set, forced: str = "auto",
           tol: float = TOLERANCE) -> str:
    """The fastest eligible candidate; RCCL when none agrees with the reference; ``RuntimeError`` when nothing is left."""
    cands = list(cands)
    good = [i for i in range(len(cands)) if off[i] < tol and cands[i] not in broken and (forced == "auto" or cands[i] == forced)]
    if not good and forced == "auto" and "rccl" in cands and "rccl" not in broken:
        good = [cands.index("rccl")]
    if not good:
        dev: Dict[str, float] = {c: float("%.3g" % v) for c, v in zip(cands, off)}
        raise RuntimeError("no usable data-parallel schedule (%s): deviations %s, timed out %s" % (forced, dev, sorted(broken)))
    return cands[min(good, key=lambda i: times[i])]
