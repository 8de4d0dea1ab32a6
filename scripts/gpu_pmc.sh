#!/bin/bash
# SQ counter pass (rocprofv3 --pmc, one pass, 8 SQ counters) over the MNIST step kernels; summary -> gpurun_out/pmc_mnist${DTF_PMC_TAG}/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_mnist${DTF_PMC_TAG}
C="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
timeout -s KILL 150 rocprofv3 --pmc $C -d gpurun_out/pmc_mnist${DTF_PMC_TAG} -o p --output-format csv -- python3 bench.py --gpus 1 --steps 48 --warmup 16 --eval 0 --extra-dtype none --ref-steps 0 --no-extras ${DTF_BENCH_ARGS} > gpurun_out/pmc_mnist${DTF_PMC_TAG}/run.log 2>&1
python scripts/pmc_summary.py $(ls gpurun_out/pmc_mnist${DTF_PMC_TAG}/*counter_collection.csv | head -1) 4 > gpurun_out/pmc_mnist${DTF_PMC_TAG}/summary.txt
rm -f gpurun_out/pmc_mnist${DTF_PMC_TAG}/*.csv
