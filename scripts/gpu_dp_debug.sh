#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out; cd $R
export DTF_HANG_DUMP_SECS=40
for i in 1 2; do
  for mode in "--no-overlap" ""; do
    start=$(date +%s)
    timeout -k 5 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo --train-size 4000 $mode > gpurun_out/dp_$i$mode.log 2>&1
    echo "run $i mode '$mode' exit $? secs $(( $(date +%s) - start ))"
    grep '"metric"' gpurun_out/dp_$i$mode.log | cut -c1-250
  done
done
