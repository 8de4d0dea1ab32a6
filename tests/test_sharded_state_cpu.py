"""State assembly of the weight-sharded data-parallel schedules (``parallel/sharded_state.py``) with two gloo ranks and a CPU
stand-in for the engine: the rows a rank does not own go stale between two ``materialize()`` calls, a checkpoint read in
between is the last agreed snapshot, a rewrite of the arenas from outside re-loads the weight buffers.  The GPU tests
(``test_rccl_gpu.py``) run the real engine; this one pins the collective logic where the driver's CPU suite sees it.
Reference behaviour being replaced: variables live whole on the parameter server (``/root/reference/demo2/train.py:27-29``).
"""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROWS, COLS, REST = 16, 4, 8   # "W_fc1" [16, 4] leads the arena, 8 more parameters follow


class _Engine:
    """What the mixin needs of an engine.  One fake step: every rank updates ITS rows of W (in the buffer of the next
    parity, on every rank: the remote stores are an all-gather here) and of M / V; the rest of the arena changes everywhere."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        n = ROWS * COLS + REST
        self.P = torch.arange(n, dtype=torch.float32) / 10
        self.M = torch.zeros(n)
        self.V = torch.zeros(n)
        self.step_t = torch.zeros(1, dtype=torch.int32)
        self.W = [torch.zeros(ROWS * COLS), torch.zeros(ROWS * COLS)]
        self.global_step = 0
        self.state_provider = None
        self.on_state_loaded = None
        self.loads = 0

    def fc1_grad_numel(self):
        return ROWS * COLS

    @staticmethod
    def shard_rows(rank, world, rows=ROWS):
        q = rows // 4
        return (q * rank // world) * 4, (q * (rank + 1) // world) * 4

    def wshard_load(self):
        self.loads += 1
        for w in self.W:
            w.copy_(self.P[:ROWS * COLS])

    def wshard_store(self):
        self.P[:ROWS * COLS].copy_(self.W[int(self.step_t.item()) & 1])

    def train_steps(self, k):
        for _ in range(k):
            c = int(self.step_t.item())
            cur, nxt = self.W[c & 1], self.W[(c + 1) & 1]
            lo, hi = self.shard_rows(self.rank, self.world)
            sl = slice(lo * COLS, hi * COLS)
            self.M[sl] += 1.0 + c
            self.V[sl] += 0.5
            mine = cur[sl] - 0.01 * self.M[sl]
            parts = [torch.empty(s[1] * COLS - s[0] * COLS) for s in (self.shard_rows(r, self.world) for r in range(self.world))]
            dist.all_gather(parts, mine)                       # "every rank stores its new rows into every rank's buffer"
            nxt.copy_(torch.cat(parts))
            self.P[ROWS * COLS:] -= 0.001                      # the replicated rest of the arena
            self.M[ROWS * COLS:] += 1.0
            self.step_t += 1
            self.global_step += 1

    def state_dict(self):
        if self.state_provider is not None:
            return self.state_provider()
        return {"P": self.P.clone(), "M": self.M.clone(), "V": self.V.clone(), "step": self.step_t.clone()}


def _worker(rank, world, initfile, outdir):
    import contextlib
    from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer
    from distributed_tensorflow_amd.parallel.sharded_state import ShardedStateMixin
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)

    class Trainer(ShardedStateMixin):
        train_steps = DataParallelTrainer.train_steps   # (the snapshot cadence lives there)

        def __init__(self, engine):
            self.engine, self.rank, self.world = engine, rank, world
            self._wsh, self._wsh_snap, self._wsh_every = False, None, 4
            self.bcasts = 0

        def _ctx(self):
            return contextlib.nullcontext()

        def _bcast(self, t, src):
            self.bcasts += 1
            dist.broadcast(t, src=src)

    e = _Engine(rank, world)
    tr = Trainer(e)
    res = {}
    tr._wsh_enter()
    res["snap0_step"] = int(e.state_dict()["step"].item())
    tr.train_steps(6)                                   # snapshot at step 4, two more steps after it
    lo, hi = e.shard_rows(rank, world)
    other = slice(0, lo * COLS) if rank else slice(hi * COLS, ROWS * COLS)
    # own rows: six updates (1 + 2 + ... + 6 = 21); the other rows: what their owner had at the snapshot step (10)
    res["others_rows_stale"] = bool((e.M[other] == 10).all()) and bool((e.M[lo * COLS:hi * COLS] == 21).all())
    res["ckpt_step"] = int(e.state_dict()["step"].item())           # what the saver thread would write now
    res["ckpt_M_whole"] = bool((e.state_dict()["M"][:ROWS * COLS] > 0).all())
    before = tr.bcasts
    tr.materialize()
    res["bcasts"] = tr.bcasts - before                              # M and V, one broadcast per owner
    tr.materialize()
    res["bcasts_again"] = tr.bcasts - before                        # nothing ran in between: nothing to assemble
    whole = torch.cat([e.P, e.M, e.V])
    parts = [torch.empty_like(whole) for _ in range(world)]
    dist.all_gather(parts, whole)
    res["identical"] = all(torch.equal(p, parts[0]) for p in parts)
    res["P_is_current_W"] = bool(torch.equal(e.P[:ROWS * COLS], e.W[int(e.step_t.item()) & 1]))
    # a restore rewrites the arenas: the buffers follow, the snapshot is the restored state
    loads = e.loads
    e.P.fill_(7.0)
    e.on_state_loaded()
    res["reloaded"] = e.loads == loads + 1 and bool((e.W[0] == 7.0).all()) and bool((e.state_dict()["P"] == 7.0).all())
    tr._wsh_leave()
    res["left"] = e.state_provider is None and e.on_state_loaded is None and not tr._wsh
    torch.save(res, os.path.join(outdir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_state_assembly_two_ranks():
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, initfile, d), nprocs=2, join=True)
        res = [torch.load(os.path.join(d, "r%d.pt" % r), weights_only=True) for r in range(2)]
    for o in res:
        assert o["snap0_step"] == 0 and o["others_rows_stale"], o
        assert o["ckpt_step"] == 4 and o["ckpt_M_whole"], o          # the agreed snapshot, whole, two steps old
        assert o["bcasts"] == 4 and o["bcasts_again"] == 4, o
        assert o["identical"] and o["P_is_current_W"] and o["reloaded"] and o["left"], o
