"""The multi-stream plan of the Inception op program (``models/inception_v3.py::StreamPlan``): pure scheduling logic, checked
on the CPU by replaying the plan with a model of stream order — every op must start after every earlier writer of its source
buffer, through the events the plan issues and nothing else."""
import random

from distributed_tensorflow_amd.models.inception_v3 import StreamPlan, build_graph


def _ordered_before(plan, ops):
    """done[i] = set of ops known to have finished when op i starts, by stream order and the plan's waits only."""
    K = plan.K
    last_on = [None] * K          # the latest op issued on each stream
    known = {}                    # op -> frozenset of ops that finished before it (transitively) plus itself
    for i in range(len(ops)):
        k = plan.stream_of[i]
        before = set()
        if last_on[k] is not None:
            before |= known[last_on[k]]
        for d in plan.waits[i]:
            assert d < i and plan.stream_of[d] != k and d in plan.waited
            before |= known[d]
        yield i, before
        known[i] = frozenset(before | {i})
        last_on[k] = i


def test_every_dependency_is_carried_by_stream_order_or_an_event():
    g = build_graph()
    rnd = random.Random(3)
    for K in (1, 2, 3, 4):
        for trial in range(3):
            costs = [rnd.uniform(5.0, 80.0) for _ in g.ops]
            plan = StreamPlan(g.ops, costs, K, device=None, edge_us=rnd.choice([0.0, 6.0, 40.0]))
            assert set(plan.stream_of) <= set(range(K)) and plan.stream_of[0] == 0
            writers = {}
            for i, before in _ordered_before(plan, g.ops):
                for d in writers.get(g.ops[i].src, []):
                    assert d in before, (K, trial, i, d)
                writers.setdefault(g.ops[i].dst, []).append(i)
            if K == 1:
                assert not plan.waited and plan.planned_us == plan.serial_us
            else:
                assert plan.planned_us < plan.serial_us          # the branches of the mixed blocks overlap in the plan
                assert plan.used and all(0 < k < K for k in plan.used)


def test_two_streams_never_need_a_relay():
    """With one side stream every cross-stream edge has the main stream at one end (the only shape the HIP runtime's
    end-of-capture survives, see StreamPlan); from three streams on, side-to-side edges exist and are relayed."""
    g = build_graph()
    costs = [20.0] * len(g.ops)
    p2 = StreamPlan(g.ops, costs, 2, device=None)
    assert all(p2.stream_of[i] == 0 or p2.stream_of[d] == 0 for i in range(len(g.ops)) for d in p2.waits[i])
    p3 = StreamPlan(g.ops, costs, 3, device=None)
    assert any(p3.stream_of[i] != 0 and p3.stream_of[d] != 0 for i in range(len(g.ops)) for d in p3.waits[i])
