set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29701
DTF_DP_SCHEDULE=ipc-shard DTF_BENCH_ARGS="--dp-path --comm ipc" bash scripts/gpu_timeline.sh shard1 || { tail -20 gpurun_out/tl_shard1/run.log; exit 1; }
tail -22 gpurun_out/tl_shard1/timeline.txt
MASTER_PORT=29702 DTF_DP_SCHEDULE=ipc-gather DTF_BENCH_ARGS="--dp-path --comm ipc" bash scripts/gpu_timeline.sh gather1 || exit 1
tail -12 gpurun_out/tl_gather1/timeline.txt
