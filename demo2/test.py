#!/usr/bin/env python
"""demo2: classify imgs/ with the newest Supervisor checkpoint in ./logs (reference: demo2/test.py,
which hard-codes logs/model.ckpt-3706)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import mnist_test  # noqa: E402

if __name__ == "__main__":
    mnist_test.main(sys.argv[1:], default_ckpt=None, default_logdir="logs")
