#!/bin/bash
# "ipc-wshard" against the other schedules: local cost on a 1-rank group, then 2 and 4 ranks sharing the GPU (auto selection
# with its trial timings, then each of wshard / shard / gather forced).
set -o pipefail
cd "$(dirname "$0")/../.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/wshard
for sch in ipc-wshard ipc-wshard2 ipc-shard ipc-gather2 ipc-gather; do
  DTF_DP_SCHEDULE=$sch MASTER_ADDR=127.0.0.1 MASTER_PORT=29733 timeout -k 10 200 python bench.py --gpus 1 --steps 3000 --warmup 300 --dtype fp32 --extra-dtype none --ref-steps 0 --no-extras --dp-path --comm ipc > gpurun_out/wshard/local-$sch.log 2>&1 || { tail -20 gpurun_out/wshard/local-$sch.log; exit 1; }
  grep "^{" gpurun_out/wshard/local-$sch.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('1-rank group  $sch %.1f us/step schedule=%s in_sync=%s' % (1000 * d['ms_per_step'], d['config'].get('dp_schedule'), d.get('replicas_in_sync')))"
done
bash scripts/gpu_dp_rehearsal.sh 2 4 || exit 1
for n in 2 4; do
  for sch in ipc-wshard ipc-wshard2 ipc-shard ipc-gather2 ipc-gather; do
    DTF_DP_SCHEDULE=$sch timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29570 + n)) \
      bench.py --gpus $n --steps 3000 --warmup 300 --backend gloo --extra-dtype none --ref-steps 0 > gpurun_out/wshard/n$n-$sch.log 2>&1 || { tail -30 gpurun_out/wshard/n$n-$sch.log; exit 1; }
    grep "^{" gpurun_out/wshard/n$n-$sch.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('N=%d shared GPU  $sch  %.1f us/step  in_sync=%s  expired_waits=%s' % (d['n_gpus'], 1000 * d['ms_per_step'], d['replicas_in_sync'], d['ipc_wait_timeouts']))"
  done
done
