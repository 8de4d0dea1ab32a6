#!/usr/bin/env python
"""demo1: single-process MNIST CNN training (reference: demo1/train.py).

    python demo1/train.py [--training_steps 10000 --batch_size 100 --data_dir MNIST_data ...]
Defaults are the reference's constants; on a machine without a GPU the eager engine runs the same
model on the CPU (BASELINE config 1, "plumbing")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import mnist_train  # noqa: E402

if __name__ == "__main__":
    sys.exit(mnist_train.main(sys.argv[1:], distributed=False))
