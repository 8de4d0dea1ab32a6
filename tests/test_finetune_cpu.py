"""Inception-v3 full fine-tune plumbing on CPU: arena layout / bucket plan, the eager trainer, 2-rank gloo data
parallelism with bucketed all-reduce, and the ``retrain2.py --finetune`` command line."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arena_layout_and_bucket_plan():
    from distributed_tensorflow_amd.models.inception_train import ArenaLayout
    from distributed_tensorflow_amd.models.inception_v3 import build_graph
    from distributed_tensorflow_amd.parallel.finetune_dp import ArenaBuckets
    g = build_graph()
    lay = ArenaLayout(g, 5)
    assert len(lay.conv_order) == 94 and lay.conv_order[0].startswith("mixed_10") and lay.conv_order[-1] == "conv"
    offs = sorted(off for off, _ in lay.entries.values())
    assert offs[0] == 0 and all(o % 64 == 0 for o in offs) and lay.size % 64 == 0
    # head first (its gradient is ready before the backward starts), stem last
    assert lay.entries["final_training_ops/weights/final_weights"][0] == 0
    assert lay.entries["conv/conv2d_params"][0] > lay.entries["mixed/conv/conv2d_params"][0] > lay.entries["mixed_9/conv/conv2d_params"][0]
    n = sum(int(torch.tensor(shape).prod()) for _o, shape in lay.entries.values())
    assert 21.7e6 < n < 21.9e6  # ~21.8 M parameters up to pool_3 + the 5-class head
    b = ArenaBuckets(lay, bucket_mb=25.0)
    assert b.ranges[0][0] == 0 and b.ranges[-1][1] == lay.size
    assert all(b.ranges[i][1] == b.ranges[i + 1][0] for i in range(len(b.ranges) - 1))
    assert 3 <= len(b.ranges) <= 5
    ends = set(lay.conv_end.values())
    assert all(e in ends for _s, e in b.ranges)
    # readiness: nothing is ready before the first conv is done; everything after the stem conv
    assert b.ready(lay.conv_end[lay.conv_order[0]], 0) == 0
    assert b.ready(lay.conv_end["conv"], 0) == len(b.ranges)
    assert len(ArenaBuckets(lay, bucket_mb=1e9).ranges) == 1


def _worker(rank, world, initfile, outdir):
    from distributed_tensorflow_amd.models.inception_train import EagerInceptionTrainer
    from distributed_tensorflow_amd.parallel.finetune_dp import FinetuneDataParallel
    torch.set_num_threads(3)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    tr = EagerInceptionTrainer(3, 2, "cpu", lr=0.05, seed=10 + rank)  # different init per rank: broadcast must fix it
    dp = FinetuneDataParallel(tr, bucket_mb=20.0)
    res = {"sync0": dp.replicas_in_sync(), "nb": len(dp.buckets.ranges)}
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 299, 299, 8, generator=g)
    y = torch.tensor([rank, 2 - rank])
    p0 = tr.P.clone()
    dp.train_step(x, y)
    res["sync1"] = dp.replicas_in_sync()
    res["moved"] = float((tr.P - p0).abs().max())
    res["gzero"] = float(tr.G.abs().max())
    res["loss"] = dp.mean_metrics()[0]
    res["P"] = tr.P[:20000].clone()
    torch.save(res, os.path.join(outdir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_keeps_replicas_identical():
    tmp = tempfile.mkdtemp()
    mp.spawn(_worker, args=(2, os.path.join(tmp, "init"), tmp), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp, "r%d.pt" % r)) for r in range(2))
    assert r0["sync0"] and r0["sync1"] and r1["sync1"]
    assert r0["nb"] >= 3
    assert r0["moved"] > 0 and r0["gzero"] == 0.0
    assert torch.equal(r0["P"], r1["P"]) and abs(r0["loss"] - r1["loss"]) < 1e-9


def test_retrain_cli_finetune_mode(tmp_path):
    cwd = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="4")
    cmd = [sys.executable, os.path.join(REPO, "retrain1", "retrain.py"), "--device", "cpu", "--synthetic", "--finetune",
           "--synthetic_classes", "2", "--synthetic_images_per_class", "12", "--training_steps", "2",
           "--finetune_batch_size", "2", "--eval_step_interval", "1", "--bottleneck_dir", os.path.join(cwd, "bn"),
           "--summaries_dir", os.path.join(cwd, "logs"), "--output_graph", os.path.join(cwd, "ft.pb"),
           "--output_labels", os.path.join(cwd, "labels.txt"), "--model_dir", os.path.join(cwd, "model"), "--seed", "3"]
    r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "Step 1: Train accuracy" in r.stdout and "Validation accuracy" in r.stdout and "Fine-tune throughput" in r.stdout
    assert "replicas in sync: True" in r.stdout
    # the export is the reference's artefact: ONE GraphDef of Inception-v3 + the new layer, which retrain*/test.py reads
    from distributed_tensorflow_amd.utils import graphdef as gd
    layers, fw, fb, final_name, ops = gd.read_retrained_graph(os.path.join(cwd, "ft.pb"))
    assert layers["conv"]["weights"].shape == (3, 3, 3, 32) and fw.shape == (2048, 2) and final_name == "final_result"
    assert sum(1 for _n, o, _i in ops if o == "Conv2D") == 94
    assert open(os.path.join(cwd, "labels.txt")).read().split("\n")[:2] == ["class 00", "class 01"]


def test_device_batch_sampler_cpu_statement():
    """The class-balanced sampler of the fine-tune (random class, then a random image of that class, with replacement:
    ``/root/reference/retrain1/retrain.py:322-340``): balanced over classes whatever the class sizes, labels match rows."""
    from distributed_tensorflow_amd.apps.finetune import DeviceBatchSampler
    y = torch.tensor([2] * 50 + [0] * 5 + [1] * 200, dtype=torch.int32)   # not label-major: sorted once
    x = torch.arange(255, dtype=torch.float32).reshape(255, 1, 1, 1).repeat(1, 1, 1, 4).contiguous()
    s = DeviceBatchSampler(x, y, 3, seed=5)
    xb, yb = torch.empty(3000, 1, 1, 4), torch.empty(3000, dtype=torch.int32)
    s.sample_into(xb, yb)
    frac = torch.bincount(yb.long(), minlength=3).float() / 3000
    assert float((frac - 1 / 3).abs().max()) < 0.04
    orig_label = y[xb[:, 0, 0, 0].long()]          # the row's value is its index in the ORIGINAL order
    assert torch.equal(orig_label, yb)
    first = xb.clone()
    s.sample_into(xb, yb)
    assert not torch.equal(first, xb)               # the draw counter advances


def test_retrain2_finetune_stop_relaunch_resume_two_ranks(tmp_path):
    """``retrain2.py --finetune`` runs under the Supervisor like every distributed script of the reference
    (``/root/reference/retrain2/retrain2.py:423-431,445,508``): a fault on the last rank stops every rank, the chief saves
    parameters + step to ./logs/model.ckpt-<step>, the relaunch restores, broadcasts, continues and exports."""
    import glob
    import re
    cwd = str(tmp_path)
    common = ["--num_gpus", "2", "--backend", "gloo", "--device", "cpu", "--synthetic", "--finetune", "--synthetic_classes", "2",
              "--synthetic_images_per_class", "16", "--finetune_batch_size", "2", "--eval_step_interval", "2",
              "--save_model_secs", "0", "--seed", "3", "--bottleneck_dir", os.path.join(cwd, "bn")]
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="3", DTF_HEARTBEAT_PERIOD_S="0.2", MASTER_PORT="29731")
    r = subprocess.run([sys.executable, os.path.join(REPO, "retrain2", "retrain2.py"), "--training_steps", "5",
                        "--fault_inject_step", "2"] + common, cwd=cwd, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode != 0, r.stdout[-2000:]
    assert "injected fault at step 2" in r.stdout and "Supervisor: final checkpoint" in r.stdout, r.stdout[-3000:]
    saved = glob.glob(os.path.join(cwd, "logs", "model.ckpt-*"))
    assert saved and not os.path.exists(os.path.join(cwd, "retrained_graph.pb"))
    step = max(int(p.rsplit("-", 1)[1]) for p in saved)
    assert 2 <= step < 5
    env["MASTER_PORT"] = "29732"
    r2 = subprocess.run([sys.executable, os.path.join(REPO, "retrain2", "retrain2.py"), "--training_steps", "5"] + common,
                        cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-3000:]
    assert "restored" in r2.stdout and ("global_step=%d" % step) in r2.stdout
    assert min(int(m) for m in re.findall(r"Step (\d+):", r2.stdout)) >= step
    assert "replicas in sync: True" in r2.stdout and os.path.exists(os.path.join(cwd, "retrained_graph.pb"))
    assert os.path.exists(os.path.join(cwd, "logs", "model.ckpt-5"))


def test_step_pairs_never_swallow_a_logging_point_or_the_fault():
    """Which steps of the fine-tune may run as two-step hipGraph replays (``apps/finetune.py``): every logging point, the
    injected fault and the last step stay single eager steps; everything else pairs up."""
    from distributed_tensorflow_amd.apps.finetune import pair_allowed
    kw = dict(start_step=0, warm_steps=5, steps=40, eval_step_interval=10, fault_inject_step=23)
    i, singles, pairs = 0, [], []
    while i < 40:
        if pair_allowed(i, **kw):
            pairs.append(i)
            i += 2
        else:
            singles.append(i)
            i += 1
    assert all(s in singles for s in (0, 1, 2, 3, 4, 10, 20, 23, 30, 39)), singles   # warm-up, logging points, fault, last
    assert all(p % 10 != 0 and (p + 1) % 10 != 0 and 23 not in (p, p + 1) for p in pairs)
    assert len(pairs) >= 12 and 2 * len(pairs) + len(singles) == 40
    # resumed run: the warm-up counts from the resumed step
    assert not pair_allowed(101, start_step=100, warm_steps=5, steps=200, eval_step_interval=10, fault_inject_step=-1)
    assert pair_allowed(105, start_step=100, warm_steps=5, steps=200, eval_step_interval=10, fault_inject_step=-1)
