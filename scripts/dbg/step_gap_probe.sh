#!/bin/bash
# Where does the gap in front of the step's first kernel come from?  Kernel traces of the fp32 loop with 16 / 1 steps per
# graph launch and without graphs, and the unprofiled step time of each.  (run through gpurun from the repo root)
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gaps
mkdir -p $OUT
B="--gpus 1 --steps 1600 --warmup 200 --extra-dtype none --ref-steps 0 --no-extras --eval 0"
for cfg in g16 g1 g64 nograph; do
  case $cfg in
    g16) export DTF_STEPS_PER_GRAPH=16; X="";;
    g1) export DTF_STEPS_PER_GRAPH=1; X="";;
    g64) export DTF_STEPS_PER_GRAPH=64; X="";;
    nograph) export DTF_STEPS_PER_GRAPH=16; X="--no-graph";;
  esac
  timeout -k 10 120 python3 bench.py $B $X > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  timeout -k 10 200 rocprofv3 --kernel-trace -d $OUT/$cfg -o tl -- python3 bench.py $B $X > $OUT/prof_$cfg.log 2>&1
  DB=$(ls $OUT/$cfg/*.db | head -1)
  python scripts/dbg/step_gaps.py $DB ${DTF_STEPS_PER_GRAPH} > $OUT/gaps_$cfg.txt
  rm -rf $OUT/$cfg
  echo "== $cfg: $(python -c "import json,sys; print(json.loads(open('$OUT/bench_$cfg.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step unprofiled"
  cat $OUT/gaps_$cfg.txt
done
