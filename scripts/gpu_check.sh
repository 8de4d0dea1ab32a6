#!/bin/bash
# GPU validation bundle: tests -> smoke -> bench (graph / no graph) -> rocprof kernel stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 400 python -u -m pytest tests -m gpu --timeout 120 --timeout-method thread -q 2>&1 | tail -40 || exit 1
timeout -k 10 120 python __graft_entry__.py --smoke || exit 1
timeout -k 10 200 python bench.py --steps 2000 --warmup 200 | tee gpurun_out/bench1.json || exit 1
timeout -k 10 200 python bench.py --steps 500 --warmup 100 --no-graph --eval 0 | tee gpurun_out/bench1_nograph.json || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -- python3 $R/bench.py --steps 300 --warmup 50 --no-graph --eval 0 > $R/gpurun_out/prof1.log 2>&1 || exit 1
ls -R $R/gpurun_out | head -30
cd $R && timeout -k 10 300 python bench.py --steps 300 --warmup 30 --engine eager --eval 0 | tee gpurun_out/bench_eager_baseline.json
