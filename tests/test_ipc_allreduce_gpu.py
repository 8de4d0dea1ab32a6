"""Direct HIP-IPC all-reduce (csrc/hip/ipc_allreduce.hip): N ranks sharing the single test GPU (RCCL refuses duplicate
devices; this backend does not care which device a peer's memory lives on)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_WORKER = r"""
import os, sys, json, time
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from distributed_tensorflow_amd.parallel.ipc_comm import IpcAllReduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 3274688  # the MNIST CNN gradient arena
comm = IpcAllReduce(n, "cuda:0", timeout_s=5.0)
ok = comm.self_test(rounds=3)
# in-place reduction of distinct per-rank data, repeated (epochs advance on the device), odd sizes
res = {"self_test": ok}
x = torch.full((n,), float(rank + 1), device="cuda")
for it in range(5):
    y = x.clone()
    comm.all_reduce(y)
    torch.cuda.synchronize()
    res["it%d" % it] = bool((y == world * (world + 1) / 2).all())
small = torch.arange(1000, dtype=torch.float32, device="cuda") * (rank + 1)
comm.all_reduce(small)
torch.cuda.synchronize()
res["small"] = bool(torch.equal(small, torch.arange(1000, dtype=torch.float32, device="cuda") * (world * (world + 1) / 2)))
# the other collectives of the backend: all-gather, broadcast, stream-ordered barrier
mine = torch.full((4096,), float(rank), device="cuda")
allg = comm.all_gather(mine)
torch.cuda.synchronize()
res["all_gather"] = bool(all(bool((allg[r] == r).all()) for r in range(world))) and tuple(allg.shape) == (world, 4096)
bc = torch.arange(2048, dtype=torch.float32, device="cuda") + 100 * rank
comm.broadcast(bc, src=world - 1)
comm.barrier()
torch.cuda.synchronize()
res["broadcast"] = bool(torch.equal(bc, torch.arange(2048, dtype=torch.float32, device="cuda") + 100 * (world - 1)))
y2 = torch.full((n,), float(rank + 1), device="cuda")
comm.all_reduce(y2)   # the two-shot all-reduce still lines up after the other collectives moved the epoch
torch.cuda.synchronize()
res["after"] = bool((y2 == world * (world + 1) / 2).all())
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(50):
    comm.all_reduce(x.clone() if False else x)  # values grow; only timing matters
torch.cuda.synchronize()
res["us_per_allreduce"] = (time.perf_counter() - t0) / 50 * 1e6
res["errors"] = comm.errors()
if rank == 0:
    print(json.dumps(res), flush=True)
dist.barrier()
comm.close()
dist.destroy_process_group()
"""


# (8 ranks, the size of one xGMI node, on request: DTF_TEST_IPC_WORLDS=2,4,8; the default keeps the suite short)
_WORLDS = [int(w) for w in os.environ.get("DTF_TEST_IPC_WORLDS", "2,4").split(",")]


@pytest.mark.parametrize("world", _WORLDS)
def test_ipc_allreduce_ranks_sharing_one_gpu(tmp_path, world):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ipc_worker.py"
    script.write_text(_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), str(script), repo]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(out)
    assert out["self_test"] is True and out["errors"] == 0 and out["small"] is True
    assert all(out["it%d" % i] for i in range(5))
    assert out["all_gather"] is True and out["broadcast"] is True and out["after"] is True


_STALL_WORKER = r"""
import os, sys, json, time
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from distributed_tensorflow_amd.parallel.ipc_comm import IpcAllReduce
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
comm = IpcAllReduce(4096, "cuda:0", timeout_s=5.0)
ok = comm.self_test(rounds=1)
comm.set_timeout(0.3)
x = torch.ones(4096, device="cuda")
t0 = time.perf_counter()
if rank == 0:            # rank 1 never joins this all-reduce: rank 0's kernels must give up, not hang
    comm.all_reduce(x)
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
res = {"self_test": ok, "errors": comm.errors(), "seconds": dt}
sys.stdout.write(json.dumps({"rank": rank, **res}) + "\n")  # (ONE write: the ranks share the pipe, print() makes two)
sys.stdout.flush()
dist.barrier()
comm.close()
dist.destroy_process_group()
"""


def test_ipc_wait_is_bounded_when_a_peer_stalls(tmp_path):
    """Failure detection: every in-kernel wait has a wall-clock bound, so a rank whose peer never shows up
    records the expiry and its grid drains (a kernel that spins forever would take the GPU with it)."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ipc_stall.py"
    script.write_text(_STALL_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), repo]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-3000:]
    outs = {}
    for l in r.stdout.splitlines():
        if l.startswith("{"):
            d = json.loads(l)
            outs[d["rank"]] = d
    assert outs[0]["self_test"] is True
    assert outs[0]["errors"] >= 1 and 0.25 < outs[0]["seconds"] < 10.0, outs
    assert outs[1]["errors"] == 0
