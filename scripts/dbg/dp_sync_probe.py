"""Which tensors differ between the replicas, and by how much, after a few data-parallel steps (ranks sharing GPU 0).
torchrun --nproc-per-node N scripts/dbg/dp_sync_probe.py [steps] [check]   (DTF_DP_SCHEDULE selects the schedule)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group
rank, world = init_process_group("gloo")
torch.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
dd = mnist.DeviceDataset(d, "cuda:0", rank=rank, world=world, seed=1)
eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype="fp32")
tr = DataParallelTrainer(eng, comm="ipc")
def report(tag):
    torch.cuda.synchronize()
    for name, arena in (("P", eng.P), ("M", eng.M), ("V", eng.V)):
        mine = arena.cpu()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        if rank == 0:
            for r in range(1, world):
                for k, (o, n, shape) in eng.layout.items():
                    diff = (allp[r][o:o + n] - allp[0][o:o + n]).abs()
                    if float(diff.max()) > 0:
                        idx = torch.nonzero(diff > 0).flatten()
                        print("%s %s: rank %d vs 0: %s differs in %d of %d elements, max %.3g, first idx %s (row %s)" % (
                            tag, name, r, k, idx.numel(), n, float(diff.max()), idx[:4].tolist(),
                            [int(i) // shape[-1] for i in idx[:4].tolist()]), flush=True)
    if rank == 0:
        print(tag, "schedule", tr.schedule, "in_sync", None, flush=True)
report("start")
if len(sys.argv) > 2 and sys.argv[2] == "quiet":  # the same without synchronising reports in between
    tr.train_steps(3)
    for k in range(2):
        tr.check_exchange()
        tr.train_steps(17)
    report("end")
elif len(sys.argv) > 2 and sys.argv[2] == "test":   # the sequence of tests/test_rccl_gpu.py::test_checked_step_...
    tr.train_steps(3); report("after 3")
    for k in range(2):
        tr.check_exchange(); report("after check %d" % k)
        tr.train_steps(17); report("after check %d + 17" % k)
else:
    tr.train_steps(1); report("after 1")
    tr.train_steps(steps); report("after %d" % (steps + 1))
    if len(sys.argv) > 2:
        print("check", tr.check_exchange()); report("after check")
        tr.train_steps(17); report("after check + 17")
print("rank", rank, "errors", tr.transport_errors(), "in_sync", tr.replicas_in_sync(), flush=True)
dist.barrier()
tr.close()
dist.destroy_process_group()
