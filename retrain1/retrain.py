#!/usr/bin/env python
"""retrain1: single-process Inception-v3 transfer learning (reference: retrain1/retrain.py).

    python retrain1/retrain.py --image_dir ./data [--training_steps 10000 --learning_rate 0.01 ...]
All 23 reference flags are accepted with their defaults.  Without ``--image_dir`` data a separable
synthetic 5-class JPEG set is generated (BASELINE config 3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.apps import retrain  # noqa: E402

if __name__ == "__main__":
    sys.exit(retrain.main(sys.argv[1:], distributed=False))
