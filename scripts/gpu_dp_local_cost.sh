#!/bin/bash
# Local cost of every data-parallel schedule: the multi-rank code path with a 1-rank group on a dedicated GPU (no xGMI
# traffic, so what is measured is the extra launches, copies and waits each schedule adds to the plain step).
# Run through gpurun from the repo root.  Optional argument: dtype (fp32 | bf16).
set -o pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
DT=${1:-fp32}
mkdir -p gpurun_out/dp_local
port=29700
for sch in plain ipc-wshard ipc-shard ipc-gather ipc-zerocopy ipc-inline ipc-overlap rccl; do
  port=$((port + 1))
  if [ $sch = plain ]; then
    extra=""
  elif [ $sch = rccl ]; then
    extra="--dp-path --comm rccl"
  else
    extra="--dp-path --comm ipc"; export DTF_DP_SCHEDULE=$sch; [ $sch = ipc-overlap ] && export DTF_IPC_FORCE_OVERLAP=1
  fi
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$port timeout -k 10 200 python bench.py --gpus 1 --steps 3000 --warmup 300 --dtype $DT --extra-dtype none --ref-steps 0 --no-extras $extra \
    > gpurun_out/dp_local/$DT-$sch.log 2>&1 || { echo "$sch FAILED"; tail -8 gpurun_out/dp_local/$DT-$sch.log; exit 1; }
  unset DTF_DP_SCHEDULE DTF_IPC_FORCE_OVERLAP
  grep "^{" gpurun_out/dp_local/$DT-$sch.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-5s %-13s %.1f us/step  schedule=%s transport=%s' % ('$DT', '$sch', 1000 * d['ms_per_step'], d['config'].get('dp_schedule'), d['config'].get('gradient_transport')))"
done
