#!/bin/bash
# One million steps of a data-parallel schedule with N ranks sharing GPU 0:  scripts/dbg/long_soak.sh <schedule> <N> [steps]
set -o pipefail
cd "$(dirname "$0")/../.."
export HSA_ENABLE_IPC_MODE_LEGACY=0 DTF_DP_SCHEDULE=$1
n=$2; steps=${3:-1000000}
mkdir -p gpurun_out/soak
# (the run prints nothing for minutes: a heartbeat file keeps the harness from taking it for hung)
( while sleep 45; do date >> gpurun_out/soak/heartbeat.txt; done ) &
hb=$!
trap 'kill $hb 2>/dev/null' EXIT
timeout -k 10 1100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29760 + n)) \
  bench.py --gpus $n --steps $steps --warmup 200 --backend gloo --extra-dtype none --ref-steps 0 > gpurun_out/soak/long_$1_n$n.log 2>&1 || { tail -20 gpurun_out/soak/long_$1_n$n.log; exit 1; }
grep "^{" gpurun_out/soak/long_$1_n$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%s N=%d  %d steps  %.1f us/step  in_sync=%s  expired_waits=%s  skipped=%s  loss=%.4f' % (d['config'].get('dp_schedule'), d['n_gpus'], d['steps'], 1000 * d['ms_per_step'], d['replicas_in_sync'], d['ipc_wait_timeouts'], d['skipped_steps'], d['final_train_loss']))"
