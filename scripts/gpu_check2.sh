#!/bin/bash
# second bundle: full GPU test suite + Inception rocprof
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 200 --timeout-method thread -q -x 2>&1 | grep -v amdgpu || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_incep -- python3 $R/scripts/bench_inception.py 64 5 nograph > $R/gpurun_out/prof_incep.log 2>&1 || exit 1
tail -2 $R/gpurun_out/prof_incep.log
