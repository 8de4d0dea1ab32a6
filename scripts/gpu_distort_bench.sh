#!/bin/bash
# Distorted retraining (reference --flip_left_right --random_crop --random_scale --random_brightness path): seconds per
# training step with the streamed pipeline vs the per-step loader.  Run through gpurun from the repo root.
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/distort
mkdir -p $OUT
for mode in 1 0; do
  W=$(mktemp -d); cd $W
  DTF_EXTRACT_PIPELINE=$mode PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 400 python -u $GRAFT_REPO_ROOT/retrain1/retrain.py --synthetic \
    --synthetic_classes 5 --synthetic_images_per_class 100 --training_steps 60 --eval_step_interval 20 --flip_left_right \
    --random_crop 10 --random_scale 10 --random_brightness 10 --seed 3 > $OUT/pipeline_$mode.log 2>&1
  echo "== DTF_EXTRACT_PIPELINE=$mode"; grep -E "Training time|distorted training stream|Final test|Head training" $OUT/pipeline_$mode.log
  cd $GRAFT_REPO_ROOT
done
