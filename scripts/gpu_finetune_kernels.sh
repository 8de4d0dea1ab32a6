#!/bin/bash
# rocprofv3 per-kernel table of the Inception-v3 fine-tune step (batch from $1, default 128) -> gpurun_out/ft_kernels/summary.txt
# Run through gpurun from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B=${1:-128}
OUT=gpurun_out/ft_kernels
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT -o ft -- python3 scripts/bench_inception_train.py --batch $B --steps 12 --warmup 3 > $OUT/run.log 2>&1
DB=$(ls $OUT/*.db | head -1)
python scripts/prof_summary.py $DB 40 0.2 > $OUT/summary.txt
grep "^{" $OUT/run.log >> $OUT/summary.txt
rm -f $DB
