"""Where two identical single-GPU fp32 runs part: max over tensors of the relative difference of Adam's first moment, step by step."""
import sys
import torch
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine

imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def run(marks):
    dd = mnist.DeviceDataset(d, "cuda:0", rank=0, world=world, seed=1)
    eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype="fp32")
    out, done = {}, 0
    for m in marks:
        eng.train_steps(m - done)
        done = m
        torch.cuda.synchronize()
        out[m] = (eng.M.clone(), eng.P.clone(), dd.perm.clone())
    return out, eng.layout


marks = list(range(1, 45))
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    a, layout = run(marks)
    b, _ = run(marks)
    line = []
    for m in marks:
        rel = max(float((a[m][0][o:o + n].double() - b[m][0][o:o + n].double()).norm() / a[m][0][o:o + n].double().norm())
                  for k, (o, n, _s) in layout.items())
        line.append("%d:%.0e%s" % (m, rel, "" if torch.equal(a[m][2], b[m][2]) else "(perm differs)"))
    print("rep %d  " % rep + " ".join(line), flush=True)
