"""Run-to-run noise of the single-GPU fp32 step: two engines, same seed, same data, same number of steps; relative difference
of Adam's first moment per tensor after 5 / 20 / 37 / 100 steps.  (What the data-parallel state tests are held against.)"""
import sys
import torch
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine

imgs, labels = mnist.synthetic_mnist(4000, seed=1)
d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)


def run(marks):
    dd = mnist.DeviceDataset(d, "cuda:0", rank=0, world=2, seed=1)
    eng = NativeMnistEngine(dd, batch_size=100, seed=5, lr=1e-4, device="cuda:0", dtype="fp32")
    out, done = {}, 0
    for m in marks:
        eng.train_steps(m - done)
        done = m
        torch.cuda.synchronize()
        out[m] = (eng.M.clone(), eng.P.clone())
    return out, eng.layout


marks = [5, 20, 37, 100]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    a, layout = run(marks)
    b, _ = run(marks)
    for m in marks:
        rel = {k: float("%.2g" % float((a[m][0][o:o + n].double() - b[m][0][o:o + n].double()).norm() / a[m][0][o:o + n].double().norm()))
               for k, (o, n, _s) in layout.items()}
        p = float((a[m][1].double() - b[m][1].double()).norm() / a[m][1].double().norm())
        print("rep %d step %3d  P %.2g  M %s" % (rep, m, p, rel), flush=True)
