#!/bin/bash
# rocprofv3 kernel timeline + per-kernel table of the MNIST step on one GPU (run through gpurun from the repo root).
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python3 bench.py --gpus 1 --steps 300 --warmup 50 > gpurun_out/tl/run.log 2>&1
DB=$(ls gpurun_out/tl/*.db | head -1)
python scripts/prof_timeline.py $DB 60 > gpurun_out/tl/timeline.txt
python scripts/prof_summary.py $DB 20 0.5 > gpurun_out/tl/summary.txt
rm -f $DB
