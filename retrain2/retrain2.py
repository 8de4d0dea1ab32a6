#!/usr/bin/env python
"""retrain2: data-parallel Inception-v3 transfer learning (reference: retrain2/retrain2.py).

Cluster flags as in the reference; bottleneck extraction is sharded over the ranks and all-gathered
into every GPU's HBM cache, the head's 41 KB gradient is all-reduced every step:

    python retrain2/retrain2.py --num_gpus 8 --image_dir ./data"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import retrain  # noqa: E402

if __name__ == "__main__":
    sys.exit(retrain.main(sys.argv[1:], distributed=True))
