#!/bin/bash
# A/B sweep of the step executor's launch-shape knobs (one bench.py run each).  Run through gpurun from the repo root.
#   scripts/gpu_sweep_knobs.sh NAME=VALUE[,NAME=VALUE] ...      e.g.  DTF_DW2_IPW=5 DTF_FC1_SPLITS=10,DTF_HEAD_BLOCKS=34
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/sweep
run() {  # name, env assignments...
  local name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 3000 --warmup 300 --no-extras --extra-dtype none --ref-steps 0 > gpurun_out/sweep/$name.log 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/sweep/$name.log; return 1; }
  grep "^{" gpurun_out/sweep/$name.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s %.2f us/step  loss %.4f' % ('$name', 1000 * d['ms_per_step'], d['final_train_loss']))"
}
run base DTF_X=0 || exit 1
for cfg in "$@"; do
  run "$cfg" ${cfg//,/ } || exit 1
done
run base_again DTF_X=0
