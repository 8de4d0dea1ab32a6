#!/bin/bash
# Same-box A/B of the multi-stream settings through the PRODUCT paths: extraction end to end, fine-tune with / without step graph.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ab
mkdir -p $OUT
for v in 1 2 1 2; do
  rm -rf /tmp/ex && mkdir /tmp/ex && cd /tmp/ex
  DTF_INCEPTION_STREAMS=$v timeout -k 10 200 python $R/retrain1/retrain.py --synthetic --synthetic_classes 5 --synthetic_images_per_class 1600 --training_steps 10 --eval_step_interval 500 --extract_batch_size 128 --seed 3 > $OUT/ex_$v.log 2>&1 || exit 1
  echo "extraction streams=$v: $(grep -o 'bottleneck extraction: .*' $OUT/ex_$v.log | cut -c1-110)"
done
ft() {  # name, env..., extra args
  rm -rf /tmp/ft && mkdir /tmp/ft && cd /tmp/ft
  timeout -k 10 200 python $R/retrain1/retrain.py --synthetic --finetune --synthetic_classes 5 --synthetic_images_per_class 200 --training_steps 150 --finetune_batch_size 128 --eval_step_interval 1000 --learning_rate 0.01 --seed 3 --finetune_bn_recalibrate 2 $2 > $OUT/ft_$1.log 2>&1 || exit 1
  echo "finetune $1: $(grep -o 'Fine-tune steady state: [0-9.]* images/sec' $OUT/ft_$1.log) $(grep -o 'step graph: .*' $OUT/ft_$1.log | cut -c1-60)"
}
DTF_FT_FWD_STREAMS=1 DTF_FT_DW_STREAM=0 ft serial_graph ""
DTF_FT_FWD_STREAMS=2 DTF_FT_DW_STREAM=1 ft streams_graph ""
DTF_FT_FWD_STREAMS=1 DTF_FT_DW_STREAM=0 ft serial_eager "--finetune_graph false"
DTF_FT_FWD_STREAMS=2 DTF_FT_DW_STREAM=1 ft streams_eager "--finetune_graph false"
DTF_FT_FWD_STREAMS=1 DTF_FT_DW_STREAM=1 ft dwonly_graph ""
