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
