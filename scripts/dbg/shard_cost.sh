set -o pipefail
bash scripts/gpu_dp_local_cost.sh fp32 || exit 1
bash scripts/gpu_dp_rehearsal.sh 2 4 || exit 1
