#!/usr/bin/env python
"""demo1: classify the digit images under imgs/ with model/train.ckpt (reference: demo1/test.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import mnist_test  # noqa: E402

if __name__ == "__main__":
    mnist_test.main(sys.argv[1:], default_ckpt="model/train.ckpt", default_logdir=None)
