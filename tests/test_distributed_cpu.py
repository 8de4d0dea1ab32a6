"""Multi-process data-parallel tests on CPU (gloo, world_size 2): no cluster needed."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import EagerMnistEngine

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split(n=256, seed=0):
    imgs, labels = mnist.synthetic_mnist(n, seed=seed)
    return mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=seed)


def _worker(rank, world, initfile, outdir, mode):
    from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer
    from distributed_tensorflow_amd.parallel.supervisor import Supervisor
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    split = _split()
    dd = mnist.DeviceDataset(split, "cpu", rank=rank, world=world, seed=11)
    eng = EagerMnistEngine(dd, batch_size=8, lr=1e-3, keep_prob=1.0, seed=100 + rank)  # different init per rank
    tr = DataParallelTrainer(eng, sync=(mode != "async"), async_period=2)
    res = {}
    if mode == "equiv":
        res["in_sync_after_broadcast"] = tr.replicas_in_sync()
        p0 = eng.P.clone()
        # explicit batches so the union is known: rank r takes rows [8r, 8r+8)
        x = torch.from_numpy(split.images[8 * rank:8 * rank + 8])
        y = torch.from_numpy(split.class_ids[8 * rank:8 * rank + 8])
        eng.fwd_bwd(3, batch=(x, y))
        dist.all_reduce(eng.G)
        eng.apply()
        res["P0"], res["P1"] = p0, eng.P.clone()
        res["in_sync_after_step"] = tr.replicas_in_sync()
    elif mode == "train":
        res["exchange_dev"] = tr.check_exchange()     # one checked step (debug assertion on the exchange itself)
        res["in_sync_after_check"] = tr.replicas_in_sync()
        # a broken exchange (rank 1's gradient never arrives in the production pass) must trip the assertion
        real = dist.all_reduce

        def lossy(t, *a, **k):
            if t.data_ptr() == eng.G.data_ptr() and rank == 1:
                t.zero_()
            return real(t, *a, **k)
        dist.all_reduce = lossy
        try:
            tr.check_exchange()
            res["caught"] = False
        except AssertionError:
            res["caught"] = True
        finally:
            dist.all_reduce = real
        tr.broadcast_state(0)
        for _ in range(4):
            tr.train_step()
        res["P"] = eng.P.clone()
        res["in_sync"] = tr.replicas_in_sync()
        res["scalars"] = tr.all_reduce_scalars([1.0, rank])
        res["steps"] = eng.global_step
    elif mode == "async":
        for _ in range(4):
            tr.train_step()
        res["in_sync"] = tr.replicas_in_sync()  # 4 % async_period == 0 -> just averaged
    elif mode == "stop":
        sv = Supervisor(tr, is_chief=(rank == 0), logdir=os.path.join(outdir, "logs"), save_model_secs=0,
                        stop_check_interval=1, log_fn=lambda m: None)
        steps = 0
        with sv.managed_session():
            while not sv.should_stop() and steps < 50:
                if rank == 1 and steps == 3:
                    sv.request_stop()
                tr.train_step()
                steps += 1
        res["steps"] = steps
    torch.save(res, os.path.join(outdir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, world=2):
    tmp = tempfile.mkdtemp()
    initfile = os.path.join(tmp, "init")
    mp.spawn(_worker, args=(world, initfile, tmp, mode), nprocs=world, join=True)
    return [torch.load(os.path.join(tmp, "r%d.pt" % r)) for r in range(world)], tmp


def test_dp2_equals_single_rank_double_batch():
    res, _ = _run("equiv")
    assert res[0]["in_sync_after_broadcast"] and res[1]["in_sync_after_broadcast"]
    assert torch.equal(res[0]["P0"], res[1]["P0"])          # rank-0 broadcast made replicas identical
    assert torch.equal(res[0]["P1"], res[1]["P1"]) and res[0]["in_sync_after_step"]
    # single process, batch = union of both shards, same starting point
    split = _split()
    dd = mnist.DeviceDataset(split, "cpu", seed=11)
    eng = EagerMnistEngine(dd, batch_size=16, lr=1e-3, keep_prob=1.0, seed=100)
    eng.P.copy_(res[0]["P0"])
    x = torch.from_numpy(split.images[:16]); y = torch.from_numpy(split.class_ids[:16])
    eng.fwd_bwd(3, batch=(x, y))
    eng.apply()
    assert torch.allclose(eng.P, res[0]["P1"], rtol=1e-4, atol=2e-6)


def test_dp2_training_keeps_replicas_identical():
    res, _ = _run("train")
    assert res[0]["in_sync"] and torch.equal(res[0]["P"], res[1]["P"])
    assert res[0]["scalars"] == [2.0, 1.0]
    assert res[0]["exchange_dev"] < 1e-5 and res[0]["in_sync_after_check"] and res[0]["steps"] == 6
    assert res[0]["caught"] and res[1]["caught"]


def test_async_emulation_averages_parameters():
    res, _ = _run("async")
    assert res[0]["in_sync"] and res[1]["in_sync"]


def test_stop_flag_propagates_and_chief_saves():
    res, tmp = _run("stop")
    assert res[0]["steps"] == res[1]["steps"] and res[0]["steps"] < 50
    from distributed_tensorflow_amd.utils import checkpoint as ckpt
    assert ckpt.latest_checkpoint(os.path.join(tmp, "logs")) is not None


def _cli(args, cwd, timeout=600):
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="2")
    return subprocess.run([sys.executable] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=timeout)


def test_demo2_cli_two_ranks_then_resume_and_infer(tmp_path):
    cwd = str(tmp_path)
    common = ["--device", "cpu", "--data_dir", os.path.join(REPO, "assets", "MNIST_data"), "--eval_step_interval", "4",
              "--eval_train_set", "false", "--seed", "1", "--batch_size", "20", "--save_model_secs", "0"]
    r = _cli([os.path.join(REPO, "demo2", "train.py"), "--num_gpus", "2", "--training_steps", "6"] + common, cwd)
    assert r.returncode == 0, r.stdout
    assert "Iter 0, test_acc:" in r.stdout and "Iter 4, test_acc:" in r.stdout and "Training time:" in r.stdout
    assert os.path.exists(os.path.join(cwd, "logs", "model.ckpt-6"))
    # relaunch with a larger budget: the chief restores ./logs/model.ckpt-6 and continues from step 6
    r2 = _cli([os.path.join(REPO, "demo2", "train.py"), "--num_gpus", "2", "--training_steps", "9"] + common, cwd)
    assert r2.returncode == 0, r2.stdout
    assert "restored" in r2.stdout and "global_step=6" in r2.stdout
    assert os.path.exists(os.path.join(cwd, "logs", "model.ckpt-9"))
    r3 = _cli([os.path.join(REPO, "demo2", "test.py"), "--imgs", os.path.join(REPO, "assets", "digit_imgs"),
               "--device", "cpu"], cwd)
    assert r3.returncode == 0, r3.stdout
    assert r3.stdout.count("识别出该图片是数字:") == 6 and "model.ckpt-9" in r3.stdout


def test_fault_injection_stops_all_ranks(tmp_path):
    cwd = str(tmp_path)
    r = _cli([os.path.join(REPO, "demo2", "train.py"), "--num_gpus", "2", "--training_steps", "40", "--device", "cpu",
              "--synthetic", "true", "--eval_step_interval", "1000", "--batch_size", "10", "--fault_inject_step", "3",
              "--save_model_secs", "0", "--seed", "2"], cwd, timeout=300)
    assert r.returncode != 0
    assert "injected fault at step 3" in r.stdout
    # the survivor (the chief) leaves its loop through the out-of-band stop flag and saves the last consistent state
    assert "stopped at global step" in r.stdout and "Supervisor: final checkpoint" in r.stdout, r.stdout[-3000:]
    import re
    names = os.listdir(os.path.join(cwd, "logs"))
    saved = [int(m.group(1)) for m in (re.fullmatch(r"model\.ckpt-(\d+)", f) for f in names) if m]
    assert saved and all(3 <= s < 40 for s in saved), names


def test_rank_that_dies_silently_ends_the_job_in_bounded_time(tmp_path):
    """The last rank disappears (os._exit) without telling anyone: the survivor is either thrown out of its collective
    by the transport or stopped by the heartbeat monitor; either way the job ends non-zero well within a minute."""
    import time
    cwd = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS="2", DTF_FAULT_KIND="die", DTF_HEARTBEAT_PERIOD_S="0.25",
               DTF_FATAL_GRACE_S="5")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "demo2", "train.py"), "--num_gpus", "2", "--training_steps",
                        "100000", "--device", "cpu", "--synthetic", "true", "--eval_step_interval", "100000",
                        "--batch_size", "10", "--fault_inject_step", "5", "--save_model_secs", "0", "--seed", "2"],
                       cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0 and "injected death at step 5" in r.stdout, r.stdout[-3000:]
    assert dt < 90, dt
