set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for sch in ipc-shard ipc-gather; do
for i in 1 2 3 4 5 6; do
  echo "=== $sch run $i"
  DTF_DP_SCHEDULE=$sch timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${2:-3} --master-addr 127.0.0.1 --master-port 29655 scripts/dbg/dp_sync_probe.py 20 quiet > gpurun_out/probe_${sch}_$i.log 2>&1 || { tail -30 gpurun_out/probe_${sch}_$i.log; exit 1; }
  grep "differs\|errors" gpurun_out/probe_${sch}_$i.log | head -12
done
done
