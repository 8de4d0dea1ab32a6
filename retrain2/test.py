#!/usr/bin/env python
"""Classify the JPEGs under imgs/ with retrained_graph.pb + retrained_labels.txt (reference: test.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import retrain_test  # noqa: E402

if __name__ == "__main__":
    retrain_test.main(sys.argv[1:])
