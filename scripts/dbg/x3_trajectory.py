"""First 200 steps of the reference schedule on the shipped digits: native fp32 (exact MFMA, bf16x6) against eager fp32
PyTorch with the same batches and dropout masks, and the two native paths against each other."""
import os
import sys
import numpy as np
import torch
from distributed_tensorflow_amd import native
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import EagerMnistEngine, NativeMnistEngine

d = os.path.join(sys.argv[1] if len(sys.argv) > 1 else "assets", "MNIST_data")
imgs = mnist.read_idx_images(os.path.join(d, "t10k-images-idx3-ubyte.gz"))
labels = mnist.read_idx_labels(os.path.join(d, "t10k-labels-idx1-ubyte.gz"))
x, y = mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels)
N = 200


def native_run(emu):
    native.hip().dtf_f32_emulation_set(emu)
    dd = mnist.DeviceDataset(mnist.DataSet(x[:8000], y[:8000], seed=11), "cuda", seed=11)
    e = NativeMnistEngine(dd, batch_size=100, seed=21, lr=1e-4, keep_prob=0.7, eval_chunk=1000, dtype="fp32")
    e.train_steps(N)
    return e.read_metrics(0, N)[0], e.seed


runs = {}
for name, emu in (("exact", 0), ("exact again", 0), ("bf16x6", 1), ("bf16x6 again", 1)):
    runs[name], seed = native_run(emu)
dd_e = mnist.DeviceDataset(mnist.DataSet(x[:8000], y[:8000], seed=11), "cuda", seed=11)
eag = EagerMnistEngine(dd_e, batch_size=100, seed=21, lr=1e-4, keep_prob=0.7, device="cuda", philox_seed=seed)
for _ in range(N):
    eag.train_step()
le = eag.read_metrics(0, N)[0]
print("max |loss difference| over the first n steps")
print("%-28s %10s %10s %10s %10s" % ("pair", "n=10", "n=30", "n=100", "n=200"))
pairs = [("exact vs eager", runs["exact"], le), ("bf16x6 vs eager", runs["bf16x6"], le),
         ("exact vs exact again", runs["exact"], runs["exact again"]), ("bf16x6 vs bf16x6 again", runs["bf16x6"], runs["bf16x6 again"]),
         ("bf16x6 vs exact", runs["bf16x6"], runs["exact"])]
for name, a, b in pairs:
    dif = np.abs(np.asarray(a) - np.asarray(b))
    print("%-28s %10.2e %10.2e %10.2e %10.2e" % (name, dif[:10].max(), dif[:30].max(), dif[:100].max(), dif[:200].max()))
