set -o pipefail
R=$PWD
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then D=$R; else D=$R/ab_old; fi
    (cd $D && timeout -k 10 120 python bench.py --gpus 1 --steps 3000 --warmup 300 --extra-dtype none --ref-steps 0 --no-extras 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v $i %.2f us' % (1000 * d['ms_per_step']))") || exit 1
  done
done
