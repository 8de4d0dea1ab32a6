"""The decision logic of the data-parallel start-up (``parallel/schedule_select.py``) with injected timings, capture
failures, expired waits and a spent budget: on the 8-GPU box a wrong branch here would otherwise be found only by the run
that was supposed to be measured (Weak 13).  What is being replaced has nothing to select:
``/root/reference/demo2/train.py:27-29`` pins the variables to the parameter server."""
import math

import pytest

from distributed_tensorflow_amd.parallel import schedule_select as sel


class FakeOps(sel.TrialOps):
    """A machine described by a table: us/step per schedule, which captures fail, whose waits expire, whose gradient is off."""

    def __init__(self, us, capture_fail=(), expire=(), wrong=(), budget_after=None, ranks_disagree=None, expire_once=()):
        self.us, self.capture_fail, self.expire, self.wrong = us, set(capture_fail), set(expire), set(wrong)
        self.expire_once = set(expire_once)   # a wait expires in the first try only (ranks time-sharing a GPU)
        self.budget_after = budget_after      # over budget once this many candidates have been used
        self.ranks_disagree = ranks_disagree  # {name: us on the slowest rank}: the max over ranks decides
        self.cur, self.used, self.restored, self.log = None, [], 0, []

    def use(self, name):
        self.cur = name
        self.used.append(name)

    def capture_ok(self):
        return self.cur not in self.capture_fail

    def first_step(self):
        if self.cur in self.expire_once:
            return 0.0                        # (the step was skipped: no gradient)
        return 2.0 if self.cur in self.wrong else 1.0

    def wait_expired(self):
        if self.cur in self.expire_once:
            self.expire_once.discard(self.cur)
            return True
        return self.cur in self.expire

    def time_steps(self):
        return self.us[self.cur]

    def restore(self):
        self.restored += 1

    def over_budget(self):
        return self.budget_after is not None and len(self.used) >= self.budget_after

    def deviation(self, g, ref):
        return abs(g - ref) / abs(ref)

    def reduce_max(self, values):
        out = list(values)
        if self.ranks_disagree:
            for i, c in enumerate(self.cands):
                if c in self.ranks_disagree and math.isfinite(out[i]):
                    out[i] = max(out[i], self.ranks_disagree[c])
        return out


ALL = dict(shard_ok=True, gather_ok=True, zc_ok=True, shared_gpu=False, allow_rccl=True, host_staged=False)
US8 = {"ipc-shard": 176.0, "ipc-gather": 210.0, "ipc-zerocopy": 212.0, "ipc-overlap": 230.0, "ipc-inline": 250.0, "rccl": 300.0}
US2 = {"ipc-shard": 240.0, "ipc-gather": 176.0, "ipc-zerocopy": 350.0, "ipc-overlap": 360.0, "ipc-inline": 380.0, "rccl": 400.0}


def _run(ops, cands, forced="auto", timed=True):
    ops.cands = cands
    times, off, broken = sel.run_trials(cands, ops, forced=forced, timed=timed, note=ops.log.append, clock=lambda: 0.0)
    return times, off, broken


def test_candidates_by_machine():
    assert sel.candidates(**ALL) == ["ipc-shard", "ipc-gather", "ipc-zerocopy", "ipc-overlap", "ipc-inline", "rccl"]
    # ranks sharing one GPU (functional runs): no forked overlap, no RCCL (it refuses duplicate devices)
    assert sel.candidates(**dict(ALL, wshard_ok=True))[:3] == ["ipc-wshard", "ipc-wshard2", "ipc-shard"]
    assert sel.candidates(**dict(ALL, wshard_ok=True, forced="ipc-wshard")) == ["ipc-wshard"]
    assert sel.candidates(**dict(ALL, wshard_ok=True, forced="ipc-wshard2")) == ["ipc-wshard2"]
    assert sel.candidates(**dict(ALL, gather2_ok=True))[:3] == ["ipc-shard", "ipc-gather2", "ipc-gather"]
    assert sel.candidates(**dict(ALL, shared_gpu=True)) == ["ipc-shard", "ipc-gather", "ipc-zerocopy", "ipc-inline"]
    assert sel.candidates(**dict(ALL, shard_ok=False, gather_ok=False, host_staged=True)) == ["ipc-zerocopy", "ipc-overlap", "ipc-inline"]
    assert sel.candidates(**dict(ALL, forced="ipc-gather")) == ["ipc-gather"]
    with pytest.raises(ValueError):
        sel.candidates(**dict(ALL, shard_ok=False, forced="ipc-shard"))
    # a forced schedule is still checked against the reference; the reference alone needs no second trial
    assert sel.trial_list(["ipc-shard"]) == ["ipc-shard", "ipc-inline"] and sel.trial_list(["ipc-inline"]) == ["ipc-inline"]


def test_fastest_valid_schedule_wins_and_it_depends_on_the_machine():
    cands = sel.candidates(**ALL)
    for us, want in ((US8, "ipc-shard"), (US2, "ipc-gather")):
        ops = FakeOps(us)
        times, off, broken = _run(ops, cands)
        assert sel.choose(cands, times, off, broken) == want
        assert ops.used == cands and ops.restored == len(cands) and not broken   # every trial starts from the same snapshot


def test_the_slowest_rank_decides():
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, ranks_disagree={"ipc-shard": 400.0})   # fast here, slow on some other rank
    times, off, broken = _run(ops, cands)
    assert sel.choose(cands, times, off, broken) == "ipc-gather"


def test_capture_failure_and_expired_wait_are_not_eligible():
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, capture_fail={"ipc-shard"}, expire={"ipc-gather"})
    times, off, broken = _run(ops, cands)
    assert broken == {"ipc-shard", "ipc-gather"} and math.isinf(times[0]) and math.isinf(times[1])
    assert sel.choose(cands, times, off, broken) == "ipc-zerocopy"
    # the schedule whose wait expired consumed a step: the snapshot is restored before the next candidate runs
    assert ops.restored == len(cands) - 1 and any("expired" in m for m in ops.log) and any("capture failed" in m for m in ops.log)


def test_wrong_gradient_is_not_eligible_even_when_fastest():
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, wrong={"ipc-shard"})
    times, off, broken = _run(ops, cands)
    assert off[0] == 1.0 and sel.choose(cands, times, off, broken) == "ipc-gather"


def test_budget_spent_only_the_reference_is_still_tried():
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, budget_after=1)
    times, off, broken = _run(ops, cands)
    assert ops.used == ["ipc-shard", "ipc-inline"]
    assert broken == {"ipc-gather", "ipc-zerocopy", "ipc-overlap", "rccl"}
    assert sel.choose(cands, times, off, broken) == "ipc-shard"


def test_rccl_is_the_last_resort_and_nothing_usable_raises():
    cands = sel.candidates(**ALL)
    every_ipc = {c for c in cands if c != "rccl"}
    ops = FakeOps(US8, wrong=every_ipc - {"ipc-inline"}, expire={"ipc-inline"})
    times, off, broken = _run(ops, cands)
    # the reference itself timed out: nothing can be validated against it, RCCL (validated by its own self-test) is what is left
    assert sel.choose(cands, times, off, broken) == "rccl"
    ops = FakeOps(US8, wrong=every_ipc - {"ipc-inline"}, expire={"ipc-inline"}, capture_fail={"rccl"})
    times, off, broken = _run(ops, cands)
    with pytest.raises(RuntimeError, match="no usable data-parallel schedule"):
        sel.choose(cands, times, off, broken)


def test_forced_schedule_must_agree_with_the_reference():
    cands = sel.trial_list(sel.candidates(**dict(ALL, forced="ipc-shard")))
    ops = FakeOps(US8)
    times, off, broken = _run(ops, cands, forced="ipc-shard")
    assert sel.choose(cands, times, off, broken, forced="ipc-shard") == "ipc-shard"
    ops = FakeOps(US8, wrong={"ipc-shard"})
    times, off, broken = _run(ops, cands, forced="ipc-shard")
    with pytest.raises(RuntimeError):
        sel.choose(cands, times, off, broken, forced="ipc-shard")


def test_untimed_trials_keep_the_order_of_preference():
    cands = sel.candidates(**dict(ALL, shared_gpu=True))
    ops = FakeOps({})
    times, off, broken = _run(ops, cands, timed=False)
    assert times == [0.0, 1.0, 2.0, 3.0] and sel.choose(cands, times, off, broken) == "ipc-shard"


def test_wait_expiring_during_the_timed_steps_drops_the_candidate():
    """TrialOps.time_steps returns inf (agreed across ranks) at the first expired wait instead of running 64 steps of 2 s
    waits: the candidate is dropped, the snapshot restored, and the remaining candidates are still tried."""
    cands = sel.candidates(**ALL)
    ops = FakeOps(dict(US8, **{"ipc-shard": math.inf}))
    times, off, broken = _run(ops, cands)
    assert "ipc-shard" in broken and math.isinf(times[cands.index("ipc-shard")])
    assert len(times) == len(off) == len(cands) and ops.used == cands and ops.restored == len(cands)
    assert sel.choose(cands, times, off, broken) == "ipc-gather"
    assert any("ipc-shard" in l and "timed steps" in l for l in ops.log)


def test_reference_whose_wait_expired_once_is_tried_again():
    """Seen with three ranks time-sharing one GPU: the reference schedule's first trial step ran into the 2 s bound, the step
    was skipped, and its zero gradient made every candidate look wrong.  The reference gets one more try from the snapshot;
    any other candidate is simply dropped."""
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, expire_once={"ipc-inline", "ipc-gather"})
    times, off, broken = _run(ops, cands)
    assert broken == {"ipc-gather"} and off[cands.index("ipc-inline")] == 0.0 and off[0] == 0.0
    assert sel.choose(cands, times, off, broken) == "ipc-shard"
    assert ops.restored == len(cands) + 1 and any("trying once more" in m for m in ops.log)
    # forced schedule, reference hiccups once: still validated and chosen
    cands = sel.trial_list(sel.candidates(**dict(ALL, forced="ipc-shard")))
    ops = FakeOps(US8, expire_once={"ipc-inline"})
    times, off, broken = _run(ops, cands, forced="ipc-shard", timed=False)
    assert not broken and sel.choose(cands, times, off, broken, forced="ipc-shard") == "ipc-shard"


def test_reference_that_keeps_expiring_is_replaced_by_the_process_group():
    """The reference schedule's exchange expired twice (three processes time-sharing one GPU): its gradient is that of a
    skipped step.  The reference gradient is then taken through the process group's all-reduce, the candidates are judged
    against it, and the reference schedule itself stays not eligible."""
    cands = sel.trial_list(sel.candidates(**dict(ALL, forced="ipc-shard")))
    ops = FakeOps(US8, expire={"ipc-inline"})
    times, off, broken = _run(ops, cands, forced="ipc-shard", timed=False)
    assert broken == {"ipc-inline"} and ops.used == ["ipc-shard", "ipc-inline", "rccl"] and off[0] == 0.0
    assert sel.choose(cands, times, off, broken, forced="ipc-shard") == "ipc-shard"
    assert any("process group" in m for m in ops.log)
    # a candidate with a wrong gradient is still caught against that reference
    ops = FakeOps(US8, expire={"ipc-inline"}, wrong={"ipc-shard"})
    times, off, broken = _run(ops, cands, forced="ipc-shard", timed=False)
    with pytest.raises(RuntimeError):
        sel.choose(cands, times, off, broken, forced="ipc-shard")
    # automatic selection with RCCL among the candidates: its trial gradient is the reference, no extra trial
    cands = sel.candidates(**ALL)
    ops = FakeOps(US8, expire={"ipc-inline"})
    times, off, broken = _run(ops, cands)
    assert ops.used == cands and sel.choose(cands, times, off, broken) == "ipc-shard"
