#!/bin/bash
# demo2 end to end on one GPU box: 2 ranks sharing GPU 0 (gloo process group + IPC gradient all-reduce), timed
# autosave / final save by the chief, resume, then inference with demo2/test.py on the shipped digit photos.
set -o pipefail
cd "$(dirname "$0")/.."
R=$PWD
W=$(mktemp -d)
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $W
timeout -k 10 300 python $R/demo2/train.py --num_gpus 2 --backend gloo --data_dir $R/assets/MNIST_data --training_steps 3000 \
  --eval_step_interval 1000 --eval_train_set false --seed 1 --check_replicas > train1.log 2>&1 || { tail -30 train1.log; exit 1; }
grep -E "Iter|Training time|Throughput|transport|IPC|saved|Saved" train1.log | tail -8
timeout -k 10 300 python $R/demo2/train.py --num_gpus 2 --backend gloo --data_dir $R/assets/MNIST_data --training_steps 4000 \
  --eval_step_interval 1000 --eval_train_set false --seed 1 > train2.log 2>&1 || { tail -30 train2.log; exit 1; }
grep -E "restored|Iter|Throughput" train2.log | tail -4
timeout -k 10 120 python $R/demo2/test.py --imgs $R/assets/digit_imgs > test.log 2>&1 || { tail -20 test.log; exit 1; }
grep -c "识别出该图片是数字" test.log
cd $R; rm -rf $W
