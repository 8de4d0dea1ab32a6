"""Where does a schedule's first-step gradient differ from the reference schedule's?  Run under torchrun with
DTF_DP_SCHEDULE=<schedule>; prints, per rank, the deviation per tensor of the arena and the rows of W_fc1 that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel import schedule_select as sel
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group

rank, world = init_process_group("gloo")
torch.cuda.set_device(0)
imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
dd = mnist.DeviceDataset(d, "cuda:0", rank=rank, world=world, seed=1)
eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype="fp32")
orig = sel.run_trials


def run_trials(cands, ops, **kw):
    od = ops.deviation

    def dev(g, ref):
        g, ref = g.detach().float().cpu(), ref.detach().float().cpu()
        for k, (o, n, _s) in eng.layout.items():
            a, b = g[o:o + n], ref[o:o + n]
            bad = int((~torch.isfinite(a)).sum())
            print("rank %d %-8s rel %.3g  nonfinite %d  max|g| %.3g max|ref| %.3g" % (
                rank, k, float((a - b).norm() / (b.norm() + 1e-30)), bad, float(a.abs().max()), float(b.abs().max())), flush=True)
            if n == 3136 * 1024:
                da = (a - b).view(3136, 1024).abs().amax(dim=1)
                rows = torch.nonzero(da > 1e-3 * float(b.abs().max())).flatten().tolist()
                cols = torch.nonzero((a - b).view(3136, 1024).abs().amax(dim=0) > 1e-3 * float(b.abs().max())).flatten().tolist()
                print("rank %d W_fc1 rows off: %d  first %s last %s; cols off %d first %s" % (
                    rank, len(rows), rows[:12], rows[-12:], len(cols), cols[:12]), flush=True)
        return od(g.to("cuda:0"), ref.to("cuda:0")) if False else float((g - ref).norm() / (ref.norm() + 1e-30))
    ops.deviation = dev
    return orig(cands, ops, **kw)


sel.run_trials = run_trials
try:
    tr = DataParallelTrainer(eng, comm="ipc")
    print("rank", rank, "schedule", tr.schedule, flush=True)
    tr.train_steps(int(os.environ.get("DBG_STEPS", "3")))
    torch.cuda.synchronize()
    tr.close()
except RuntimeError as e:
    print("rank", rank, "error:", e, flush=True)
dist.barrier()
dist.destroy_process_group()
