#!/bin/bash
# End-to-end result collection on one MI355X: the numbers quoted in README.md / BASELINE.md §4.
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/results
mkdir -p $OUT
W=$(mktemp -d)
export HSA_ENABLE_IPC_MODE_LEGACY=0
# demo1: MNIST CNN on the shipped t10k split (8000 train / 1000 val / 1000 test), reference hyper-parameters
timeout -k 10 300 python demo1/train.py --data_dir assets/MNIST_data --training_steps 3000 --eval_step_interval 500 --log_dir $W/logs --checkpoint_path $W/model/train.ckpt > $OUT/demo1_train.log 2>&1 || exit 1
tail -4 $OUT/demo1_train.log
(cd $W && timeout -k 10 120 python $OLDPWD/demo1/test.py --imgs $OLDPWD/assets/digit_imgs --checkpoint $W/model/train.ckpt) > $OUT/demo1_test.log 2>&1 || exit 1
# headline bench, 1 GPU
timeout -k 10 200 python bench.py --steps 3000 --warmup 300 > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err || exit 1
cat $OUT/bench_1gpu.json
# retrain1: Inception-v3 bottleneck extraction + head retrain on a synthetic 5-class set
timeout -k 10 500 python retrain1/retrain.py --synthetic --synthetic_classes 5 --synthetic_images_per_class 200 --training_steps 2000 \
  --eval_step_interval 500 --bottleneck_dir $W/bn --summaries_dir $W/rl --output_graph $W/rg.pb --output_labels $W/labels.txt \
  --model_dir $W/model_dir > $OUT/retrain1.log 2>&1 || exit 1
grep -E "extraction|Final test|throughput|Training time" $OUT/retrain1.log
# full fine-tune (BASELINE config 5), B=64 and B=128 per GPU + PyTorch eager baseline
timeout -k 10 300 python scripts/bench_inception_train.py --batch 64 --steps 8 --warmup 3 --eager > $OUT/finetune_b64.log 2>&1 || exit 1
timeout -k 10 300 python scripts/bench_inception_train.py --batch 128 --steps 6 --warmup 2 > $OUT/finetune_b128.log 2>&1 || exit 1
grep -h "^{" $OUT/finetune_b64.log $OUT/finetune_b128.log
rm -rf $W
