#!/bin/bash
# Fine-tune through the product path with and without the two-step hipGraph (apps/finetune.py: StepPairGraph).
set -o pipefail
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out/ftgraph
for g in true false; do
  W=$(mktemp -d); cd $W
  timeout -k 10 280 python $R/retrain1/retrain.py --synthetic --finetune --synthetic_classes 5 --synthetic_images_per_class 200 \
    --training_steps 150 --finetune_batch_size 128 --eval_step_interval 50 --finetune_bn_recalibrate 2 --finetune_graph $g \
    > $R/gpurun_out/ftgraph/graph_$g.log 2>&1 || { tail -30 $R/gpurun_out/ftgraph/graph_$g.log; exit 1; }
  cd $R; rm -rf $W
  echo "finetune_graph=$g"; grep -E "step graph|steady state|Step [0-9]+:|Final test" gpurun_out/ftgraph/graph_$g.log | tail -8
done
