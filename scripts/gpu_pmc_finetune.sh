#!/bin/bash
# SQ counter pass over the Inception-v3 fine-tune step (batch from $1, default 128); summary -> gpurun_out/pmc_ft/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
B=${1:-128}
mkdir -p gpurun_out/pmc_ft
C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout -s KILL 280 rocprofv3 --pmc $C -d gpurun_out/pmc_ft -o p --output-format csv -- python3 scripts/bench_inception_train.py --batch $B --steps 4 --warmup 2 > gpurun_out/pmc_ft/run.log 2>&1
python scripts/pmc_summary.py $(ls gpurun_out/pmc_ft/*counter_collection.csv | head -1) 4 > gpurun_out/pmc_ft/summary.txt
rm -f gpurun_out/pmc_ft/*.csv
