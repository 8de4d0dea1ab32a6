"""distributed_tensorflow_amd — an MI355X-native (gfx950 / CDNA4) training framework.

Capabilities mirror the BonneyBB/distributed_tensorflow tutorial repo (MNIST CNN
train/test, async-PS -> synchronous RCCL data parallelism, Inception-v3 bottleneck
retraining) but the architecture is MI355X-first:

* hot path   : hand-written HIP kernels (``csrc/hip``) — MFMA implicit-GEMM conv,
               fused bias+ReLU+max-pool epilogues, split-K dense GEMMs, fused
               double-softmax cross-entropy head, multi-tensor Adam/SGD;
* runtime    : a native C++ step executor that launches the whole train step and
               replays it as a hipGraph (``csrc/hip/mnist_engine.hip``);
* scale-out  : one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL)
               bucketed gradient all-reduce over xGMI overlapped with backward;
* host side  : native C++ tfevents writer / CRC32C / TCP rendezvous store /
               checkpoint container (``csrc/host``).

Sub-packages: ``data`` ``models`` ``ops`` ``optim`` ``parallel`` ``engine`` ``utils``.
"""

__version__ = "0.1.0"

from . import flags  # noqa: F401
