set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -u -m pytest tests/test_ipc_allreduce_gpu.py tests/test_engine_gpu.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
for n in 2 4; do
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 1000 --warmup 96 --backend gloo > gpurun_out/b$n.log 2>&1 || { tail -20 gpurun_out/b$n.log; exit 1; }
grep "^{" gpurun_out/b$n.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['ms_per_step'], d['value'], d['replicas_in_sync'], d['ipc_wait_timeouts'], d['allreduce_us_at_startup'])"
done
