#!/bin/bash
# Functional rehearsal of the multi-GPU bench on a ONE-GPU box: N ranks share GPU 0 (gloo bootstrap, because RCCL refuses
# duplicate devices; the gradient exchange itself is the IPC path that real multi-GPU runs use).  Prints, per N, the
# chosen schedule, the trial timings, replica consistency and expired waits.  Run through gpurun from the repo root.
set -o pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/rehearsal
for n in ${@:-1 2 4}; do
  if [ "$n" = 1 ]; then
    timeout -k 10 200 python bench.py --gpus 1 --steps 2000 --warmup 200 > gpurun_out/rehearsal/n1.log 2>&1 || { tail -20 gpurun_out/rehearsal/n1.log; exit 1; }
  else
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29560 + n)) \
      bench.py --gpus $n --steps 2000 --warmup 200 --backend gloo > gpurun_out/rehearsal/n$n.log 2>&1 || { tail -30 gpurun_out/rehearsal/n$n.log; exit 1; }
  fi
  grep "^{" gpurun_out/rehearsal/n$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('N=%d  %.1f us/step  %.0f images/s  schedule=%s  trials=%s  in_sync=%s  expired_waits=%s' % (
    d['n_gpus'], 1000 * d['ms_per_step'], d['value'], d['config'].get('dp_schedule'), d.get('dp_schedule_us_at_startup'),
    d['replicas_in_sync'], d['ipc_wait_timeouts']))"
done
