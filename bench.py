#!/usr/bin/env python
"""Headline benchmark: training images/sec of the demo2 MNIST CNN under synchronous data
parallelism (BASELINE.json: "demo2 MNIST CNN sync data-parallel ... on 8xMI355X").

    python bench.py --gpus N --steps K --warmup W

N == 1 runs in-process; N > 1 is launched by the driver as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU
over RCCL); when started directly with N > 1 it re-launches itself that way as a child process.

Config (reference constants): per-GPU batch 100 (``demo1/train.py:9``; weak scaling: global batch
100*N), Adam 1e-4, dropout keep 0.7, double-softmax cross-entropy, bf16 MFMA operands with fp32
accumulation and fp32 master weights.  The full step is timed: on-device batch gather, forward,
loss head, backward, gradient all-reduce, fused Adam.  Data is synthetic MNIST-shaped (the
reference tree ships without train-images) and weights are random-init.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

BASELINE_IMAGES_PER_SEC = 620.0  # indirect hint from demo2/test.py:182 (ckpt-3706 / 600 s autosave)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--batch", type=int, default=100, help="per-GPU batch (reference: 100)")
    p.add_argument("--engine", type=str, default="native")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-overlap", action="store_true")
    p.add_argument("--backend", type=str, default="auto", help="auto (nccl==RCCL on GPUs) | gloo (functional tests)")
    p.add_argument("--comm", type=str, default="auto", choices=["auto", "ipc", "rccl"],
                   help="gradient transport: direct IPC/xGMI all-reduce, RCCL collectives, or auto (ipc if its self-test passes)")
    p.add_argument("--train-size", type=int, default=55000)
    p.add_argument("--dp-path", action="store_true",
                   help="diagnostic: with --gpus 1, create a 1-rank RCCL group and take the multi-rank step path")
    p.add_argument("--eval", type=int, default=1, help="report final test accuracy on a held-out synthetic split")
    return p.parse_args()


def main():
    args = parse()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    if os.environ.get("DTF_HANG_DUMP_SECS"):  # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DTF_HANG_DUMP_SECS"]), exit=True)

    import torch
    import torch.distributed as dist

    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.engine.mnist_engine import make_engine
    from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer, init_process_group

    rank, world = init_process_group(args.backend) if args.gpus > 1 else (0, 1)
    if args.dp_path and args.gpus == 1 and torch.cuda.is_available():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        local = int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        device = "cuda:%d" % local
    else:
        device = "cpu"
        args.engine = "eager"

    # same synthetic data on every rank; each rank trains on its own rank-strided shard
    imgs, labels = mnist.synthetic_mnist(args.train_size, seed=1)
    train = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
    dd = mnist.DeviceDataset(train, device, rank=rank, world=world, seed=1)
    eng = make_engine(args.engine, dd, batch_size=args.batch, lr=1e-4, keep_prob=0.7, double_softmax=True,
                      seed=1234, device=device, use_graph=not args.no_graph)
    trainer = DataParallelTrainer(eng, sync=True, overlap=not args.no_overlap, force_collectives=args.dp_path, comm=args.comm)

    def sync():
        if world > 1:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    trainer.prepare()  # graph capture only: nothing executes
    trainer.train_steps(args.warmup)
    sync()
    if world > 1:
        trainer.ensure_healthy()  # (IPC waits timed out during warm-up -> every rank switches to RCCL before the clock starts)
    t0 = time.perf_counter()
    trainer.train_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    test_acc = None
    last_loss = None
    if args.eval:
        ti, tl = mnist.synthetic_mnist(10000, seed=2)
        test = mnist.DataSet(mnist._scale_flatten(ti), mnist.dense_to_one_hot(tl), seed=2)
        tdd = mnist.DeviceDataset(test, device, seed=2)
        _loss, test_acc = eng.evaluate(tdd.images, tdd.labels)
        total = eng.global_step
        lm, _am = eng.read_metrics(max(0, total - 50), total)
        last_loss = float(lm.mean())
    in_sync = trainer.replicas_in_sync()

    if rank == 0:
        global_batch = args.batch * world
        value = args.steps * global_batch / dt
        out = {
            "metric": "training images/sec (MNIST CNN, sync data-parallel)",
            "value": round(value, 1),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt / args.steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": round(value / BASELINE_IMAGES_PER_SEC, 2),
            "dtype": "bf16" if args.engine == "native" else "fp32",
            "data": "synthetic MNIST-shaped (28x28x1, 10 classes), random-init weights",
            "config": {"model": "demo2 MNIST CNN (conv5x5x32-pool-conv5x5x64-pool-fc1024-dropout-fc10), Adam 1e-4",
                       "global_batch": global_batch, "seq_len": None, "parallelism": "dp%d" % world,
                       "per_gpu_batch": args.batch, "engine": args.engine, "hipgraph": not args.no_graph,
                       "overlap_allreduce": (not args.no_overlap) if trainer._multi else None,
                       "gradient_transport": trainer.comm if trainer._multi else None,
                       "dp_schedule": getattr(trainer, "schedule", None)},
            "final_test_accuracy": None if test_acc is None else round(float(test_acc), 4),
            "final_train_loss": last_loss,
            "replicas_in_sync": in_sync,
            "dp_schedule_us_at_startup": getattr(trainer, "schedule_us", None),
            "ipc_wait_timeouts": trainer.transport_errors() if trainer.ipc is not None else None,
            "allreduce_us_at_startup": trainer.transport_us,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    trainer.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
