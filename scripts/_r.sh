set -e
timeout -k 10 300 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5
for c in 8 1 16 8; do
  DTF_STEPS_PER_GRAPH=$c timeout -k 10 120 python bench.py --gpus 1 --steps 3000 --warmup 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk=$c', d['ms_per_step'], d['value'], d['final_test_accuracy'], d['final_train_loss'])"
done
