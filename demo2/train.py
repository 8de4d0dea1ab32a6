#!/usr/bin/env python
"""demo2: distributed MNIST CNN training (reference: demo2/train.py).

Same flags as the reference (--ps_hosts --worker_hosts --job_name --task_index); the
asynchronous gRPC parameter server is replaced by synchronous RCCL all-reduce data parallelism:

    # one process per GPU on one node
    python demo2/train.py --num_gpus 8
    # or explicitly, reference style
    python demo2/train.py --job_name=worker --task_index=0 --worker_hosts=hostA:2223,hostB:2224
    python demo2/train.py --job_name=ps --ps_hosts=hostC:2222       # optional: hosts the rendezvous store
    # or under torchrun (RANK/WORLD_SIZE from the environment)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import mnist_train  # noqa: E402

if __name__ == "__main__":
    sys.exit(mnist_train.main(sys.argv[1:], distributed=True))
