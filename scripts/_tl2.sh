set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/tl2
export WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 LOCAL_WORLD_SIZE=2
RANK=1 LOCAL_RANK=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 400 --warmup 56 --backend gloo > gpurun_out/tl2/rank1.log 2>&1 &
P1=$!
export RANK=0 LOCAL_RANK=0
timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/tl2 -o tl -- python3 bench.py --gpus 2 --steps 400 --warmup 56 --backend gloo > gpurun_out/tl2/rank0.log 2>&1
wait $P1
DB=$(ls gpurun_out/tl2/*.db | head -1)
python scripts/prof_timeline.py $DB 120 > gpurun_out/tl2/timeline.txt
rm -f $DB
grep "^{" gpurun_out/tl2/rank0.log | cut -c1-200
