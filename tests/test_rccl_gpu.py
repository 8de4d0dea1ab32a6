"""Native RCCL communicator (``csrc/hip/rccl_comm.hip``) and the survivability of the data-parallel step (GPU only).

One test box has one GPU, so the communicator is exercised as a 1-rank group (the full code path: unique id through our
rendezvous store, ncclCommInitRank, collectives captured into a hipGraph) and the multi-rank protocols with 2 ranks
sharing the GPU over the IPC exchange.  Reference behaviour being replaced: ``/root/reference/demo2/train.py:13-29,166-184``.
"""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_communicator_single_rank_collectives_and_graph_capture():
    from distributed_tensorflow_amd import native
    from distributed_tensorflow_amd.parallel.cluster import StoreClient, StoreServer
    from distributed_tensorflow_amd.parallel.rccl_comm import RcclCommunicator
    assert RcclCommunicator.available()
    server = StoreServer(0)
    client = StoreClient("127.0.0.1", server.port)
    comm = RcclCommunicator.create(0, 1, "cuda:0", store=client, key="rccl/test")
    assert client.get("rccl/test", timeout_s=1.0) is not None and comm.world == 1 and comm.version >= 20000
    x = torch.arange(4096, dtype=torch.float32, device="cuda")
    y = x.clone()
    comm.all_reduce(y)
    comm.all_reduce_multi([y[:1024], y[1024:]])
    comm.broadcast(y, root=0)
    g = comm.all_gather(y)
    torch.cuda.synchronize()
    assert torch.equal(y, x) and g.shape == (1, 4096) and torch.equal(g[0], x)
    assert comm.async_error() is None
    # the collective is a plain enqueue on the caller's stream: captured into a hipGraph and replayed
    lib = native.hip()
    st = torch.cuda.Stream()
    buf = torch.ones(1 << 20, device="cuda")
    st.synchronize()
    import ctypes as C
    native.check(lib.dtf_graph_begin(st.cuda_stream), "begin")
    with torch.cuda.stream(st):
        buf.mul_(2.0)
        comm.all_reduce(buf, stream=st)
        buf.add_(1.0)
    h = C.c_void_p()
    native.check(lib.dtf_graph_end(st.cuda_stream, C.byref(h)), "end")
    assert float(buf[0].item()) == 1.0  # capturing records, it does not execute
    for _ in range(3):
        native.check(lib.dtf_graph_launch(h, st.cuda_stream), "launch")
    st.synchronize()
    assert float(buf[0].item()) == 15.0 and float(buf[-1].item()) == 15.0  # ((1*2+1)*2+1)*2+1
    lib.dtf_graph_free(h)
    comm.close()
    client.close()
    server.stop()


def _run_bench(extra, env_extra=None, nproc=1, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", env["MASTER_PORT"], os.path.join(REPO, "bench.py"), "--gpus", str(nproc)]
    else:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"]
    r = subprocess.run(cmd + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r.stdout


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_rccl_schedule_is_one_graph_launch_single_rank_group(dtype):
    """``bench.py --dp-path``: a 1-rank RCCL group takes the multi-rank code path; with the native communicator the
    collectives are inside the step graph (the JSON says so) and the step costs little more than the plain one."""
    out, log = _run_bench(["--dp-path", "--comm", "rccl", "--steps", "600", "--warmup", "100", "--dtype", dtype,
                           "--extra-dtype", "none", "--ref-steps", "0", "--train-size", "8000"],
                          {"MASTER_PORT": "29641"})
    assert out["config"]["gradient_transport"] == "rccl" and out["config"]["dp_schedule"] == "rccl"
    assert out["rccl_in_step_graph"] is True and out["rccl_nranks"] == 1 and out["dtype"] == dtype
    assert out["dp_startup_seconds"] is not None and out["dp_startup_seconds"] < 5.0, log[-2000:]
    assert out["replicas_in_sync"] is True and out["final_train_loss"] < 2.35
    assert out["ms_per_step"] < (0.130 if dtype == "bf16" else 0.200), out["ms_per_step"]


_STALLED_PEER = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group
rank, world = init_process_group("gloo")
torch.cuda.set_device(0)
imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
dd = mnist.DeviceDataset(d, "cuda:0", rank=rank, world=world, seed=1)
eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0",
                        dtype="fp32" if os.environ["DTF_DP_SCHEDULE"] == "ipc-wshard" else "bf16")
tr = DataParallelTrainer(eng, comm="ipc")
tr.prepare()
tr.train_steps(5)
tr.sync_params()   # (a schedule that keeps W_fc1 outside the arena brings P up to date; a no-op for the others)
torch.cuda.synchronize()
P0, step0 = eng.P.clone(), int(eng.step_t.item())
dist.barrier()
if rank == 1:
    time.sleep(1.5)   # a peer that stalls for longer than the exchange's bound (0.4 s)
tr.train_step()
tr.sync_params()
torch.cuda.synchronize()
out = {"rank": rank, "skipped": eng.skipped_steps(), "unchanged": bool(torch.equal(eng.P, P0)),
       "step_before": step0, "step_after": int(eng.step_t.item()), "errors": tr.transport_errors(),
       "grad_discarded": float(eng.G[eng.layout["b_fc1"][0]:].abs().max().item()) == 0.0}
dist.barrier()
if os.environ.get("DTF_TEST_AFTER") == "reset":
    # what the start-up does after dropping a candidate schedule whose wait expired: forget the expiry on every rank
    for c in (tr.ipc, tr.ipc_act, tr.ipc_w):
        if c is not None:
            c.reset_errors()
    dist.barrier()
out["healthy"] = tr.ensure_healthy()            # collective: agrees on the failure, falls back, re-broadcasts the state
out["transport_after"] = tr.comm
tr.train_steps(5)
tr.sync_params()
torch.cuda.synchronize()
out["skipped_after"], out["errors_after"] = eng.skipped_steps(), tr.transport_errors()
out["in_sync"] = tr.replicas_in_sync()
out["host_step"], out["device_step"] = eng.global_step, int(eng.step_t.item())
json.dump(out, open(os.path.join(sys.argv[2], "r%d.json" % rank), "w"))  # (two ranks' stdout lines can interleave)
dist.barrier()
tr.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("schedule", ["ipc-inline", "ipc-wshard"])
def test_expired_ipc_wait_skips_the_optimizer_on_every_rank(tmp_path, schedule):
    """A peer that stalls past the in-kernel bound: the waiting rank's exchange times out, NO rank applies that step
    (parameters bit-identical to before, step counter not advanced, gradient discarded), the trainer then falls back to the
    process group and training continues with replicas in sync (VERDICT r1 'Next round' 2c).  Under "ipc-wshard" the
    fallback first assembles the sharded optimizer state."""
    script = tmp_path / "stalled_peer.py"
    script.write_text(_STALLED_PEER)
    env = dict(os.environ, MASTER_PORT="29643", HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE=schedule,
               DTF_IPC_TIMEOUT_S="0.4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29643", str(script), REPO, str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    res = [json.load(open(str(tmp_path / ("r%d.json" % k)))) for k in range(2)]
    for o in res:
        assert o["skipped"] == 1 and o["unchanged"] and o["step_after"] == o["step_before"] and o["grad_discarded"], o
        assert o["healthy"] is False and o["transport_after"] == "rccl" and o["in_sync"] is True, o
        assert o["host_step"] == o["device_step"] == o["step_before"] + 5, o
    assert max(o["errors"] for o in res) >= 1 and min(o["errors"] for o in res) == 0, res


def test_reference_schedule_that_keeps_expiring_is_replaced_by_the_process_group(tmp_path):
    """Fault injection (``DTF_DP_FAULT_INJECT=reference-expires``): the reference schedule's trial step "expires" twice, as
    seen with three processes time-sharing one GPU.  The forced schedule is then validated against the process group's
    all-reduce of the same gradients instead of being refused, and training runs on it with the replicas in sync."""
    script = tmp_path / "sharded_state.py"
    script.write_text(_SHARDED_STATE)
    env = dict(os.environ, MASTER_PORT="29657", HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE="ipc-shard", DTF_TEST_TAG="a",
               DTF_DP_FAULT_INJECT="reference-expires")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29657", str(script), REPO, str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "taking the reference gradient through the process group" in r.stdout, r.stdout[-3000:]
    res = [json.load(open(str(tmp_path / ("a_r%d.json" % q)))) for q in range(2)]
    for o in res:
        assert o["schedule"] == "ipc-shard" and o["in_sync"] and o["errors"] == 0 and o["live_step"] == 37, o
        assert o["live"] == res[0]["live"], o


def test_cleared_expiry_count_lets_the_ipc_schedule_continue_without_a_spurious_skip(tmp_path):
    """``IpcAllReduce.reset_errors()`` (start-up, after a candidate schedule with an expired wait was dropped): this rank's
    count, the copy the peers read and the total each rank has acted on are cleared together, so the steps that follow are
    applied on every rank (a stale published count would differ from the total acted on and skip one more step on one
    rank), the trainer stays on the IPC transport and the replicas stay in sync."""
    script = tmp_path / "stalled_peer.py"
    script.write_text(_STALLED_PEER)
    env = dict(os.environ, MASTER_PORT="29644", HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE="ipc-inline",
               DTF_IPC_TIMEOUT_S="0.4", DTF_TEST_AFTER="reset")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29644", str(script), REPO, str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    res = [json.load(open(str(tmp_path / ("r%d.json" % k)))) for k in range(2)]
    assert max(o["errors"] for o in res) >= 1, res   # (the stalled step did expire somewhere)
    for o in res:
        assert o["skipped"] == 1 and o["unchanged"], o
        assert o["healthy"] is True and o["transport_after"] == "ipc", o
        assert o["skipped_after"] == 1 and o["errors_after"] == 0 and o["in_sync"] is True, o
        # (the host's count includes the skipped step; the start-up restores it with the snapshot, `ensure_healthy()` on its
        # fallback path re-reads it from the device)
        assert o["device_step"] == o["step_before"] + 5 and o["host_step"] == o["device_step"] + 1, o


_CHECKED = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group
rank, world = init_process_group("gloo")
torch.cuda.set_device(0)
imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
dd = mnist.DeviceDataset(d, "cuda:0", rank=rank, world=world, seed=1)
eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype=sys.argv[2])
tr = DataParallelTrainer(eng, comm="ipc")
out = {"rank": rank, "schedule": tr.schedule, "dev": []}
tr.train_steps(3)
for _ in range(2):
    out["dev"].append(tr.check_exchange())
    tr.train_steps(17)
out["in_sync"] = tr.replicas_in_sync()
out["errors"] = tr.transport_errors()
if not out["in_sync"]:  # diagnostics: which tensors differ from rank 0's, in how many elements, by how much
    mine = eng.P.cpu()
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine)
    out["diff"] = {k: [int((allp[r][o:o + n] != allp[0][o:o + n]).sum()) for r in range(world)] + [float((allp[rank][o:o + n] - allp[0][o:o + n]).abs().max())]
                   for k, (o, n, _s) in eng.layout.items()}
    out["finite"] = bool(torch.isfinite(mine).all())
out["host_step"], out["device_step"] = eng.global_step, int(eng.step_t.item())
lm, am = eng.read_metrics(0, eng.global_step)
out["loss_first"], out["loss_last"] = float(lm[:5].mean()), float(lm[-5:].mean())
json.dump(out, open(os.path.join(sys.argv[3], "r%d.json" % rank), "w"))
dist.barrier()
tr.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("schedule,dtype,nproc", [("ipc-inline", "fp32", 2), ("ipc-zerocopy", "fp32", 2), ("ipc-gather", "fp32", 2),
                                                  ("ipc-gather2", "fp32", 2), ("ipc-gather2", "fp32", 3),
                                                  ("ipc-shard", "fp32", 2), ("ipc-shard", "fp32", 3),
                                                  ("ipc-wshard", "fp32", 2), ("ipc-wshard", "fp32", 3),
                                                  ("ipc-wshard2", "fp32", 2), ("ipc-wshard2", "fp32", 3),
                                                  ("ipc-gather", "bf16", 2), ("ipc-inline", "bf16", 2)] +
                         # (eight ranks on the one GPU -- four K sets in the slice GEMM, 392-row slices: slow, on request)
                         ([("ipc-wshard", "fp32", 8), ("ipc-wshard2", "fp32", 8), ("ipc-shard", "fp32", 8)]
                          if os.environ.get("DTF_TEST_DP_WORLD8") == "1" else []))
def test_checked_step_validates_the_production_exchange(tmp_path, schedule, dtype, nproc):
    """``check_exchange`` (the ``--check_replicas`` assertion): the gradient the optimizer consumed under the production
    schedule equals the host-ordered sum of the ranks' local gradients; the checked step leaves counters, metrics and
    replicas consistent."""
    script = tmp_path / "checked.py"
    script.write_text(_CHECKED)
    env = dict(os.environ, MASTER_PORT="29644", HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE=schedule)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", "29644", str(script), REPO, dtype, str(tmp_path)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    res = [json.load(open(str(tmp_path / ("r%d.json" % k)))) for k in range(nproc)]
    for o in res:
        assert o["schedule"] == schedule and max(o["dev"]) < 2e-3, o
        assert o["in_sync"] and o["host_step"] == o["device_step"] == 3 + 2 * 18, o
        assert 1.5 < o["loss_last"] < o["loss_first"] + 0.05 < 2.6, o   # (the checked steps did not double-count metrics)


_SHARDED_STATE = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group
rank, world = init_process_group("gloo")
torch.cuda.set_device(0)
imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
dd = mnist.DeviceDataset(d, "cuda:0", rank=rank, world=world, seed=1)
eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype="fp32")
tr = DataParallelTrainer(eng, comm="ipc")
sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()
sched = os.environ["DTF_TEST_TAG"]
out = {"rank": rank, "schedule": tr.schedule}
def live(tag):
    tr.materialize()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"P": eng.P.cpu(), "M": eng.M.cpu(), "V": eng.V.cpu(), "layout": {k: (o, n) for k, (o, n, _s) in eng.layout.items()}},
                   os.path.join(sys.argv[2], "%s_%s.pt" % (sched, tag)))
    return [sha(t) for t in (eng.P, eng.M, eng.V)]
tr.train_steps(5)
out["live5"] = live("5")
tr.train_steps(18)
tr.train_steps(14)                      # 37 steps: snapshots at 16 and 32 (DTF_WSHARD_SNAPSHOT_STEPS=16)
sd = eng.state_dict()                   # what a checkpoint written NOW would hold
out["ckpt_step"] = int(sd["step"].item())
out["ckpt"] = [sha(sd[k]) for k in ("P", "M", "V")]
out["live"] = live("37")
out["live_step"] = int(eng.step_t.item())
out["in_sync"] = tr.replicas_in_sync()
out["errors"] = tr.transport_errors()
if len(sys.argv) > 3 and sys.argv[3] == "resume":
    # load the checkpoint state into the live engine (every rank: same state) and run to step 37 again
    eng.load_state_dict(sd)
    tr.train_steps(37 - out["ckpt_step"])
    out["resumed"] = live("resumed")
    # once more from the same snapshot: what two runs of the SAME five steps differ by (atomics reorder, ReLU / argmax flips)
    eng.load_state_dict(sd)
    tr.train_steps(37 - out["ckpt_step"])
    live("resumed2")
json.dump(out, open(os.path.join(sys.argv[2], "%s_r%d.json" % (sched, rank)), "w"))
dist.barrier()
tr.close()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("nproc", [2, 3])
def test_weight_sharded_schedule_keeps_the_same_state_as_the_gradient_sharded_one(tmp_path, nproc):
    """"ipc-wshard" (updated weight rows travel, W_fc1's optimizer state is sharded) against "ipc-shard" (gradient rows
    travel, every rank updates everything): the same row gradients go through the same Adam arithmetic, so after
    ``materialize()`` P, M and V agree between the two schedules up to the run-to-run noise of the step (the conv filter
    gradients' fp32 atomics, see the thresholds below), and bit for bit across the ranks of a run.  A checkpoint taken between two snapshot steps holds the state of the last one (consistent, a few steps
    old), and resuming from it leads back to the live state."""
    import torch
    script = tmp_path / "sharded_state.py"
    script.write_text(_SHARDED_STATE)
    res = {}
    # two runs of the gradient-sharded schedule give the noise floor (the conv filter gradients' atomics reorder from run
    # to run and early training amplifies that), the third run is the weight-sharded schedule
    for k, (tag, schedule) in enumerate((("a", "ipc-shard"), ("b", "ipc-shard"), ("w", "ipc-wshard"))):
        port = str(29651 + k)
        env = dict(os.environ, MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE=schedule, DTF_TEST_TAG=tag,
                   DTF_WSHARD_SNAPSHOT_STEPS="16")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", port, str(script), REPO, str(tmp_path)] + (["resume"] if tag == "w" else [])
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-3000:]
        res[tag] = [json.load(open(str(tmp_path / ("%s_r%d.json" % (tag, q))))) for q in range(nproc)]

    def rel(x, y):
        a, b = (torch.load(str(tmp_path / ("%s.pt" % n)), weights_only=True) for n in (x, y))
        return {t: float((a[t].double() - b[t].double()).norm() / a[t].double().norm()) for t in ("P", "M", "V")}

    def rel_by_tensor(x, y, t="M"):
        a, b = (torch.load(str(tmp_path / ("%s.pt" % n)), weights_only=True) for n in (x, y))
        return {k: float("%.2g" % float((a[t][o:o + n].double() - b[t][o:o + n].double()).norm() / a[t][o:o + n].double().norm()))
                for k, (o, n) in a["layout"].items()}

    by_tensor, by_tensor_resumed = rel_by_tensor("a_5", "w_5"), rel_by_tensor("w_37", "w_resumed")
    resumed_floor = rel_by_tensor("w_resumed", "w_resumed2")
    print("first moment by tensor after 5 steps: floor", rel_by_tensor("a_5", "b_5"), "wshard vs shard", by_tensor,
          "resumed vs live", by_tensor_resumed, "resumed vs resumed again", resumed_floor)
    # Two runs of the same steps part in discrete events (scripts/dbg/noise_floor2.py, single GPU, no data parallelism: 3e-7
    # for the first steps, then jumps to 4e-5 around step 10, 3e-4 .. 2e-3 around step 30: the filter gradients' fp32 atomics
    # reorder, and a ReLU / max-pool decision on a knife edge then falls the other way).  A weight buffer that is one step
    # stale anywhere (forward, data gradient, update) shows as >= 1e-3 within five steps; optimizer rows that did not come
    # back from a checkpoint as >= 0.1.
    assert max(by_tensor.values()) < 3e-4, by_tensor
    assert max(by_tensor_resumed.values()) < 1e-2, (by_tensor_resumed, resumed_floor)
    floor5, floor37 = rel("a_5", "b_5"), rel("a_37", "b_37")
    early, late, again = rel("a_5", "w_5"), rel("a_37", "w_37"), rel("w_37", "w_resumed")
    print("noise floor after 5 / 37 steps", floor5, floor37, "wshard vs shard", early, late, "resumed vs live", again)
    for t in ("P", "M", "V"):
        assert early[t] < 3e-4, (t, early)
        assert late[t] < max(50 * floor37[t], 1e-2), (t, late, floor37)   # (sanity only: 37 steps amplify the noise)
    for tag in ("a", "b", "w"):
        for o in res[tag]:
            assert o["in_sync"] and o["errors"] == 0 and o["live_step"] == 37, o
            assert o["live"] == res[tag][0]["live"] and o["live5"] == res[tag][0]["live5"], o   # same bits on every rank of a run
    for o in res["a"]:
        assert o["schedule"] == "ipc-shard" and o["ckpt_step"] == 37 and o["ckpt"] == o["live"], o
    for o in res["w"]:
        assert o["schedule"] == "ipc-wshard" and o["ckpt_step"] == 32, o       # the last agreed snapshot
        assert o["ckpt"] == res["w"][0]["ckpt"] and o["ckpt"] != o["live"], o
        assert o["resumed"] == res["w"][0]["resumed"], o                        # (bit-identical across the ranks of the run)


_HEAD_RCCL = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29646", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
from distributed_tensorflow_amd.models.retrain_head import RetrainHead
from distributed_tensorflow_amd.parallel.rccl_comm import RcclCommunicator
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(2)
N, C, B = 400, 5, 100
cache = torch.rand(N, 2048, device="cuda"); labels = torch.randint(0, C, (N,), device="cuda", dtype=torch.int32)
a, b = RetrainHead(C, "cuda", learning_rate=0.05, seed=1), RetrainHead(C, "cuda", learning_rate=0.05, seed=1)
a.comm = RcclCommunicator.create(0, 1, "cuda:0", key="rccl/headtest")   # the data-parallel path with a 1-rank group
calls = {"n": 0}
orig = dist.all_reduce
def counting(*x, **k):
    calls["n"] += 1
    return orig(*x, **k)
dist.all_reduce = counting
import time
for i in range(300):
    rows = torch.randint(0, N, (B,), device="cuda", dtype=torch.int32)
    a.train_step(cache, labels, rows); b.train_step(cache, labels, rows)
torch.cuda.synchronize()
maxdiff = float((a.P - b.P).abs().max())
t0 = time.perf_counter()
for i in range(300):
    a.train_step(cache, labels, rows)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 300 * 1e6
dist.all_reduce = orig
print(json.dumps({"pg_calls": calls["n"], "maxdiff": maxdiff, "us_per_dp_step": us,
                  "async_error": a.comm.async_error()}))
a.comm.close()
dist.destroy_process_group()
"""


def test_retrain_head_dp_step_uses_the_native_communicator(tmp_path):
    """retrain2's head step with the framework's communicator attached: forward, backward, ncclAllReduce of the 41 KB
    gradient and the SGD update are four enqueues on one stream -- no torch.distributed collective, no host sync; same
    parameters as the fused single-GPU step (``/root/reference/retrain2/retrain2.py:377,411-416``)."""
    script = tmp_path / "head_rccl.py"
    script.write_text(_HEAD_RCCL)
    r = subprocess.run([sys.executable, str(script), REPO], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["pg_calls"] == 0 and out["maxdiff"] < 1e-6 and out["async_error"] is None, out
    assert out["us_per_dp_step"] < 150.0, out


def test_demo2_two_ranks_weight_sharded_schedule_final_checkpoint_and_resume(tmp_path):
    """demo2 with two ranks sharing the GPU under "ipc-wshard" (sharded optimizer state): the checked steps pass, the final
    checkpoint is the state of the LAST step (assembled collectively at the clean end of training, not the last snapshot),
    and a relaunch restores it, re-loads W_fc1's buffers and trains on (``/root/reference/demo2/train.py:166-184``)."""
    cwd = str(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DTF_DP_SCHEDULE="ipc-wshard", DTF_WSHARD_SNAPSHOT_STEPS="64",
               PYTHONPATH=REPO)
    common = ["--num_gpus", "2", "--backend", "gloo", "--synthetic", "true", "--eval_step_interval", "100", "--eval_train_set",
              "false", "--seed", "3", "--check_replicas"]
    # (the chief's saver thread fires while training runs: what it writes is the last agreed snapshot, named after ITS step)
    r = subprocess.run([sys.executable, os.path.join(REPO, "demo2", "train.py"), "--training_steps", "301", "--save_model_secs",
                        "1"] + common, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "schedule ipc-wshard" in r.stdout and r.stdout.count("checked step") >= 2, r.stdout[-3000:]
    assert os.path.exists(os.path.join(cwd, "logs", "model.ckpt-301")), os.listdir(os.path.join(cwd, "logs"))
    auto = [int(m) for m in re.findall(r"autosave checkpoint \S*model\.ckpt-(\d+)", r.stdout)]
    # (a snapshot step: a multiple of 64, the step after a checked step's state assembly, or the start)
    assert all(a <= 301 for a in auto), auto
    common = common + ["--save_model_secs", "0"]
    acc1 = [float(m) for m in re.findall(r"test_acc: ([0-9.]+)", r.stdout)]
    r2 = subprocess.run([sys.executable, os.path.join(REPO, "demo2", "train.py"), "--training_steps", "402"] + common, cwd=cwd, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r2.returncode == 0, r2.stdout[-3000:]
    assert "restored" in r2.stdout and "global_step=301" in r2.stdout, r2.stdout[-3000:]
    acc2 = [float(m) for m in re.findall(r"test_acc: ([0-9.]+)", r2.stdout)]
    assert acc1[-1] > 0.6 and acc2 and acc2[-1] >= acc1[-1] - 0.05, (acc1, acc2)   # (it continued: a fresh start is at 0.1)
