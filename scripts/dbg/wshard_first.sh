#!/bin/bash
# First GPU contact of the "ipc-wshard" schedule: kernel numerics, 2- and 3-rank checked steps, sharded-state test, local cost.
set -o pipefail
cd "$(dirname "$0")/../.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/wshard
timeout -k 10 300 python -u -m pytest tests/test_f32_gpu.py -k "trio_f32_shard" -x -q --timeout 120 --timeout-method thread > gpurun_out/wshard/t1.log 2>&1 || { tail -30 gpurun_out/wshard/t1.log; exit 1; }
tail -2 gpurun_out/wshard/t1.log
timeout -k 10 500 python -u -m pytest tests/test_rccl_gpu.py -k "wshard or weight_sharded" -x -q --timeout 200 --timeout-method thread > gpurun_out/wshard/t2.log 2>&1 || { tail -60 gpurun_out/wshard/t2.log; exit 1; }
tail -2 gpurun_out/wshard/t2.log
for sch in ipc-wshard ipc-shard; do
  DTF_DP_SCHEDULE=$sch MASTER_ADDR=127.0.0.1 MASTER_PORT=29733 timeout -k 10 200 python bench.py --gpus 1 --steps 3000 --warmup 300 --dtype fp32 --extra-dtype none --ref-steps 0 --no-extras --dp-path --comm ipc > gpurun_out/wshard/local-$sch.log 2>&1 || { tail -20 gpurun_out/wshard/local-$sch.log; exit 1; }
  grep "^{" gpurun_out/wshard/local-$sch.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$sch %.1f us/step schedule=%s in_sync=%s' % (1000 * d['ms_per_step'], d['config'].get('dp_schedule'), d.get('replicas_in_sync')))"
done
