set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -u -m pytest tests/test_rccl_gpu.py -k "checked_step and shard" --timeout 200 --timeout-method thread -x -q > gpurun_out/shard_t2.log 2>&1 || { tail -60 gpurun_out/shard_t2.log; exit 1; }
tail -2 gpurun_out/shard_t2.log
timeout -k 10 170 bash scripts/gpu_dp_rehearsal.sh 2 4 || { grep -v "^\[W\|Gloo\|amdgpu.ids" gpurun_out/rehearsal/n4.log | tail; exit 1; }
MASTER_PORT=29701 DTF_DP_SCHEDULE=ipc-shard DTF_BENCH_ARGS="--dp-path --comm ipc" bash scripts/gpu_timeline.sh shard1 || { tail -20 gpurun_out/tl_shard1/run.log; exit 1; }
head -12 gpurun_out/tl_shard1/summary.txt | cut -c1-60,110-150
grep "^{" gpurun_out/tl_shard1/run.log | cut -c1-200
