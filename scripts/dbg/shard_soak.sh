set -o pipefail
export DTF_DP_SCHEDULE=ipc-shard
bash scripts/gpu_dp_soak.sh 2 4
