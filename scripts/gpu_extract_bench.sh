#!/bin/bash
# End-to-end bottleneck extraction rate of retrain1/retrain.py --synthetic on the 320x240 synthetic set.
# usage (through gpurun, from the repo root): bash scripts/gpu_extract_bench.sh [images_per_class] [batch]
set -e
cd $GRAFT_REPO_ROOT
N=${1:-1600}
B=${2:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/extract
mkdir -p $OUT
W=$(mktemp -d)
cd $W
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 500 python -u $GRAFT_REPO_ROOT/retrain1/retrain.py --synthetic --synthetic_classes 5 \
  --synthetic_images_per_class $N --training_steps 20 --eval_step_interval 10 --extract_batch_size $B --seed 3 > $OUT/run_b$B.log 2>&1
grep -E "extraction|Final test" $OUT/run_b$B.log
