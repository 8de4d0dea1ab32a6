#!/bin/bash
# rocprofv3 kernel timeline + per-kernel table of the MNIST step on one GPU (run through gpurun from the repo root).
# Optional argument: a tag for the output directory (gpurun_out/tl_<tag>); knobs come from the environment.
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/tl${1:+_$1}
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT -o tl -- python3 bench.py --gpus 1 --steps 300 --warmup 50 --extra-dtype none --ref-steps 0 --no-extras ${DTF_BENCH_ARGS} > $OUT/run.log 2>&1
DB=$(ls $OUT/*.db | head -1)
python scripts/prof_timeline.py $DB 60 > $OUT/timeline.txt
python scripts/prof_summary.py $DB 20 0.5 > $OUT/summary.txt
rm -f $DB
