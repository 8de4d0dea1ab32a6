#!/bin/bash
# Soak of the data-parallel step on a ONE-GPU box: 100 000 steps with 2 and 4 ranks sharing GPU 0 (gloo bootstrap, IPC
# exchange); prints schedule, replica consistency and expired waits.  DTF_DP_SCHEDULE=ipc-gather etc. forces a schedule.
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/soak
for n in ${@:-2 4}; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29660 + n)) \
    bench.py --gpus $n --steps 100000 --warmup 200 --backend gloo --extra-dtype none --ref-steps 0 ${DTF_BENCH_ARGS} > gpurun_out/soak/n$n.log 2>&1 || { tail -20 gpurun_out/soak/n$n.log; exit 1; }
  grep "^{" gpurun_out/soak/n$n.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('N=%d  %d steps  %.1f us/step  %.0f images/s  schedule=%s  in_sync=%s  expired_waits=%s  final_test_accuracy=%s  loss=%.4f' % (
    d['n_gpus'], d['steps'], 1000 * d['ms_per_step'], d['value'], d['config'].get('dp_schedule'), d['replicas_in_sync'], d['ipc_wait_timeouts'], d['final_test_accuracy'], d['final_train_loss']))"
done
