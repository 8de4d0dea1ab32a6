"""``bench.py`` contract (one JSON line, the keys and semantics the driver reads) on CPU: eager engine, tiny sizes, one
process and two gloo ranks launched exactly as the driver launches them."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def _check(d: dict, n: int, steps: int, warmup: int) -> None:
    assert REQUIRED <= set(d), sorted(REQUIRED - set(d))
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup     # echo the CLI
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["unit"] == "images/s"
    cfg = d["config"]
    assert {"model", "global_batch", "seq_len", "parallelism"} <= set(cfg)
    assert cfg["global_batch"] == 100 * n and cfg["parallelism"] == "dp%d" % n
    # value is the whole-job aggregate: global batch per step time
    assert abs(d["value"] - cfg["global_batch"] / (d["ms_per_step"] / 1e3)) <= 0.01 * d["value"] + 0.1
    assert d["dtype"] == "fp32" and "synthetic" in d["data"]
    assert d["vs_baseline"] is None or abs(d["vs_baseline"] - d["value"] / 620.0) < 0.02 * d["vs_baseline"] + 0.01


def test_bench_json_contract_single_process():
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--engine", "eager", "--steps", "2", "--warmup", "1",
                        "--train-size", "400", "--ref-steps", "0", "--extra-dtype", "none"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    d = _line(r.stdout)
    _check(d, 1, 2, 1)
    assert d["config"]["hipgraph"] is False and d["config"]["engine"] == "eager"


def test_bench_json_contract_two_ranks_as_the_driver_launches_it():
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(REPO, "bench.py"), "--gpus", "2", "--engine", "eager",
                        "--backend", "gloo", "--steps", "2", "--warmup", "1", "--train-size", "800", "--ref-steps", "0",
                        "--extra-dtype", "none"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    d = _line(r.stdout)
    _check(d, 2, 2, 1)
    assert d["replicas_in_sync"] is True


def test_extras_records_schema_from_driver_logs():
    """The labelled extras of bench.py (BASELINE configs 3-5) are parsed from the product drivers' own log lines."""
    from distributed_tensorflow_amd.apps import bench_extras as be
    log = ("bottleneck extraction: 2000 images on this rank in 0.21s (9523.8 images/sec/rank, HBM cache 16.4 MB)\n"
           "2026-01-01: Step 1999: Train accuracy = 100.0% Cross entropy = 0.91 Validation accuracy = 100.0%\n"
           "Final test accuracy = 99.5% \nTraining time: 0.11s\n"
           "Head training throughput: 1840000.0 bottleneck-images/sec (global batch 100)\n")
    p = be.parse_retrain1(log)
    assert p["extraction_images_per_sec"] == 9523.8 and p["extraction_images"] == 2000 and p["head_images_per_sec"] == 1840000.0
    assert abs(p["final_test_accuracy"] - 0.995) < 1e-9 and p["training_seconds"] == 0.11
    log = ("2026: Step 0: Train accuracy = 20.0% Cross entropy = 1.6094 Validation accuracy = 21.0%\n"
           "Training time: 1.23s\nFine-tune throughput: 5200.0 images/sec (global batch 128, 1 GPU; batch sampling on the device)\n"
           "Fine-tune steady state: 6100.5 images/sec (40 steps after 5 warm-up steps, sampling, validation and logging included; "
           "last train loss 0.0123)\nFinal test accuracy = 100.0% (N=60)\n")
    p = be.parse_finetune(log)
    assert p["value"] == 6100.5 and p["timed_steps"] == 40 and p["final_train_loss"] == 0.0123 and p["first_train_loss"] == 1.6094
    assert p["whole_run_images_per_sec"] == 5200.0 and p["final_test_accuracy"] == 1.0
    # nothing runs without a GPU budget: an empty selection returns no records
    assert be.run_extras(budget_s=0.0) == {}
