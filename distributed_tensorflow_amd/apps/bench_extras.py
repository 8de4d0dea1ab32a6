"""Labelled extras of ``bench.py``: the other BASELINE.json configurations, measured through the PRODUCT paths (the
command-line drivers a user runs), each in its own process with its own time limit so that a failure cannot touch the
headline number.  Nothing here is the headline; every record says what it is.

* ``extra_retrain1`` — config 3 (``/root/reference/retrain1/retrain.py:372-476``): synthetic 5-class image folders, JPEG
  files -> batched Inception-v3 forward -> HBM bottleneck cache (images/s end to end), head training (steps/s), final
  test accuracy;
* ``extra_inception_forward`` — the batched Inception-v3 forward alone at batch 128 (bf16 operands, hipGraph);
* ``extra_finetune_bf16`` (and ``extra_finetune_bf16_b256``) — config 5: the full fine-tune at batch 128 (256) measured THROUGH ``retrain1/retrain.py --finetune``
  (device-side sampling, logging and validation included), with the loss after the run.
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile
import time
from typing import Dict, Optional

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _run(cmd, cwd: str, timeout_s: float):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        return r.returncode, r.stdout, time.perf_counter() - t0
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        return -9, out + "\n(timed out after %.0f s)" % timeout_s, time.perf_counter() - t0


def _num(pattern: str, text: str) -> Optional[float]:
    m = re.findall(pattern, text)
    return float(m[-1]) if m else None


def parse_retrain1(out: str) -> Dict[str, object]:
    return {"extraction_images_per_sec": _num(r"bottleneck extraction: \d+ images on this rank in [\d.]+s \(([\d.]+) images/sec", out),
            "extraction_images": _num(r"bottleneck extraction: (\d+) images", out),
            "head_images_per_sec": _num(r"Head training throughput: ([\d.]+) bottleneck-images/sec", out),
            "final_test_accuracy": (lambda v: None if v is None else v / 100.0)(_num(r"Final test accuracy = ([\d.]+)%", out)),
            "training_seconds": _num(r"Training time: ([\d.]+)s", out)}


def parse_finetune(out: str) -> Dict[str, object]:
    return {"value": _num(r"Fine-tune steady state: ([\d.]+) images/sec", out),
            "whole_run_images_per_sec": _num(r"Fine-tune throughput: ([\d.]+) images/sec", out),
            "timed_steps": _num(r"Fine-tune steady state: [\d.]+ images/sec \((\d+) steps", out),
            "final_train_loss": _num(r"last train loss ([\d.]+)", out),
            "first_train_loss": (lambda m: float(m[0]) if m else None)(re.findall(r"Step 0: Train accuracy = [\d.]+% Cross entropy = ([\d.]+)", out)),
            "final_test_accuracy": (lambda v: None if v is None else v / 100.0)(_num(r"Final test accuracy = ([\d.]+)%", out))}


def run_extras(budget_s: float = 130.0, which=("retrain1", "inception_forward", "finetune_bf16")) -> Dict[str, dict]:
    t_begin = time.perf_counter()
    out: Dict[str, dict] = {}
    py = sys.executable

    def left() -> float:
        return budget_s - (time.perf_counter() - t_begin)

    with tempfile.TemporaryDirectory(prefix="dtf_bench_") as tmp:
        if "inception_forward" in which and left() > 20:
            rc, log, dt = _run([py, os.path.join(ROOT, "scripts", "bench_inception.py"), "128", "20"], tmp, min(60.0, left()))
            rec = {"config": "Inception-v3 (2015 graph) forward to pool_3, batch 128, random-init weights, synthetic input",
                   "dtype": "bf16", "unit": "images/s", "wall_seconds": round(dt, 1)}
            lines = [l for l in log.splitlines() if l.startswith("{")]
            if rc == 0 and lines:
                j = json.loads(lines[-1])
                rec.update(value=j["images_per_sec"], ms=j["ms_per_batch"], steps=20, model_tflops=j["tflops"], hipgraph=j["hipgraph"])
            else:
                rec.update(value=None, error=log[-400:])
            out["extra_inception_forward"] = rec
        if "retrain1" in which and left() > 30:
            steps = 2000
            rc, log, dt = _run([py, os.path.join(ROOT, "retrain1", "retrain.py"), "--synthetic", "--synthetic_classes", "5",
                                "--synthetic_images_per_class", "1600", "--training_steps", str(steps), "--eval_step_interval", "500",
                                "--extract_batch_size", "128", "--seed", "3"], tmp, min(90.0, left()))
            rec = {"config": "retrain1: 5 synthetic classes x 1600 JPEG files -> Inception-v3 bottlenecks (HBM cache) -> "
                             "softmax head, SGD 0.01, batch 100 train / 100 validation per step, double softmax",
                   "dtype": "bf16 backbone / fp32 head", "unit": "images/s", "steps": steps, "wall_seconds": round(dt, 1)}
            p = parse_retrain1(log)
            if rc == 0 and p["extraction_images_per_sec"]:
                rec.update(value=p["extraction_images_per_sec"], what="JPEG files -> HBM bottleneck cache, end to end",
                           ms=None if not p["extraction_images"] else round(1000.0 * p["extraction_images"] / p["extraction_images_per_sec"], 1),
                           head_bottleneck_images_per_sec=p["head_images_per_sec"],
                           head_steps_per_sec=None if not p["head_images_per_sec"] else round(p["head_images_per_sec"] / 100.0, 1),
                           final_test_accuracy=p["final_test_accuracy"])
            else:
                rec.update(value=None, error=log[-400:])
            out["extra_retrain1"] = rec
        # batch 128 (the batch the earlier rounds quote) and batch 256 ("large global batch": 16.7 GB of activations of the
        # 288 GB, the 8x8 and 17x17 layers fill the 256 CUs only there)
        for batch, steps, key in ((128, 150, "extra_finetune_bf16"), (256, 100, "extra_finetune_bf16_b256")):
            if "finetune_bf16" not in which or left() <= 45:
                continue
            rc, log, dt = _run([py, os.path.join(ROOT, "retrain1", "retrain.py"), "--synthetic", "--finetune", "--synthetic_classes", "5",
                                "--synthetic_images_per_class", "200", "--training_steps", str(steps),
                                "--finetune_batch_size", str(batch), "--eval_step_interval", "1000", "--learning_rate", "0.01",
                                "--seed", "3", "--finetune_bn_recalibrate", "2"], tmp, min(110.0, left()))
            rec = {"config": "Inception-v3 full fine-tune through retrain1/retrain.py --finetune: batch %d, SGD 0.01, 5 synthetic "
                             "classes, batch drawn by the device-side sampler, random-init weights" % batch,
                   "dtype": "bf16", "unit": "images/s", "steps": steps, "wall_seconds": round(dt, 1)}
            p = parse_finetune(log)
            if rc == 0 and p["value"]:
                rec.update(p)
                rec["ms"] = round(1000.0 * batch / p["value"], 2)
            else:
                rec.update(value=None, error=log[-400:])
            out[key] = rec
    return out
