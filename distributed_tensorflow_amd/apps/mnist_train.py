"""Training driver shared by ``demo1/train.py`` (single process) and ``demo2/train.py``
(cluster flags -> synchronous data parallelism).

Reference loops: ``demo1/train.py:149-165`` and ``demo2/train.py:176-193``:
10 000 steps of batch 100 with keep_prob 0.7; ``cross_entropy``/``accuracy`` summaries every step;
every 100 steps accuracy on the 10 000 test images and on the whole train split, printed as
``Iter N, test_acc: X, train_acc = Y``; ``Training time: %3.2fs``; checkpoint at the end
(demo1: ``model/train.ckpt``; demo2: Supervisor autosave to ``./logs/model.ckpt-<step>``).

Differences by design: per-step summaries are buffered on the device and
flushed in batches (Q5); full-split evaluation is chunked on the device and sharded over ranks
(Q4); only rank 0 touches the filesystem (Q9); the global step counts synchronous steps of
global batch ``100 x world`` (Q6); a final checkpoint is always written (Q8); ``--job_name=ps``
hosts the rendezvous store and exits when the workers are done (Q12).
"""
from __future__ import annotations

import os
import sys
import time
from typing import List, Optional

import numpy as np


def _log(rank: int, msg: str) -> None:
    if rank == 0:
        print(msg, flush=True)


def _flush_step_summaries(writer, engine, first: int, last: int) -> None:
    from ..utils.summary import Summary
    if writer is None or last <= first:
        return
    loss, acc = engine.read_metrics(first, last)
    for i, step in enumerate(range(first, last)):
        writer.add_summary(Summary().scalar("cross_entropy/cross_entropy", loss[i]).scalar("acc/accuracy/accuracy", acc[i]), step)


def _sharded_eval(trainer, images, labels):
    """(loss, accuracy) over a device split; each rank evaluates a contiguous 1/world slice."""
    n = images.shape[0]
    world, rank = trainer.world, trainer.rank
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    if hi > lo:
        loss, acc = trainer.engine.evaluate(images[lo:hi], labels[lo:hi])
        sums = [loss * (hi - lo), acc * (hi - lo), float(hi - lo)]
    else:
        sums = [0.0, 0.0, 0.0]
    tot = trainer.all_reduce_scalars(sums)
    return tot[0] / tot[2], tot[1] / tot[2]


def run(FLAGS, distributed: bool) -> int:
    import torch

    from ..data import mnist
    from ..engine.mnist_engine import make_engine
    from ..parallel import cluster as cl
    from ..parallel.ddp import DataParallelTrainer, init_process_group
    from ..parallel.supervisor import ConsistencyError, Supervisor
    from ..utils import checkpoint as ckpt
    from ..utils.summary import FileWriter, Summary, variable_summaries

    rank, world, server = 0, 1, None
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed and under_launcher:
        rank, world = init_process_group(FLAGS.backend)
    elif distributed:
        if FLAGS.num_gpus and FLAGS.num_gpus > 1:
            from ..parallel.launcher import spawn_local_workers
            return spawn_local_workers(sys.argv[0], [a for a in sys.argv[1:] if not a.startswith("--num_gpus")],
                                       FLAGS.num_gpus)
        spec = cl.ClusterSpec.from_flags(FLAGS.ps_hosts, FLAGS.worker_hosts)
        server = cl.Server(spec, FLAGS.job_name, FLAGS.task_index, backend=FLAGS.backend)
        if FLAGS.job_name == "ps":
            server.join()
            return 0
        rank, world = server.init_collectives()
    device = cl.bind_device(rank, FLAGS.device)
    if FLAGS.seed is not None:
        torch.manual_seed(FLAGS.seed)
        np.random.seed(FLAGS.seed)

    data = mnist.read_data_sets(FLAGS.data_dir, one_hot=True, synthetic=FLAGS.synthetic, seed=FLAGS.seed)
    _log(rank, "MNIST source: %s (train %d / validation %d / test %d)" % (
        data.source, data.train.num_examples, data.validation.num_examples, data.test.num_examples))
    shard_seed = FLAGS.seed if FLAGS.seed is not None else 20240601  # same permutation on every rank -> disjoint shards
    train_dd = mnist.DeviceDataset(data.train, device, rank=rank, world=world, seed=shard_seed)
    test_dd = mnist.DeviceDataset(data.test, device, seed=shard_seed)

    kind = FLAGS.engine
    if kind == "auto":
        kind = "native" if device.type == "cuda" else "eager"
    engine = make_engine(kind, train_dd, batch_size=FLAGS.batch_size, lr=FLAGS.learning_rate, keep_prob=FLAGS.keep_prob,
                         double_softmax=FLAGS.double_softmax, seed=FLAGS.seed, device=device, use_graph=FLAGS.use_graph,
                         dtype=FLAGS.dtype)
    trainer = DataParallelTrainer(engine, sync=FLAGS.sync, async_period=FLAGS.async_period,
                                  comm=getattr(FLAGS, "comm", "auto"))
    _log(rank, "engine=%s dtype=%s device=%s world=%d global_batch=%d sync=%s" % (
        kind, FLAGS.dtype if kind == "native" else "fp32", device, world, FLAGS.batch_size * world, FLAGS.sync))

    writer = None
    if rank == 0:
        graph = {"model": "mnist_cnn", "engine": kind, "world": world,
                 "step_kernels": ["conv1_fwd", "conv2_fwd_pool", "fc1_fwd_splitk", "head",
                                  "fc_dw_pair|fc1_bwd_data", "unpool2+conv2_bwd_data", "conv2_dw|conv1_dw", "adam_tf"]}
        writer = FileWriter(FLAGS.log_dir, graph)

    sv: Optional[Supervisor] = None
    store = store_server = None
    if distributed:
        # the stop flag and the heartbeats travel out of band, through the job's rendezvous store
        if world > 1:
            store = server.connect_store() if server is not None else None
            if store is None:
                store, store_server = cl.job_store(rank, world)
        sv = Supervisor(trainer, is_chief=(rank == 0), logdir=FLAGS.log_dir, save_model_secs=FLAGS.save_model_secs,
                        summary_writer=writer, log_fn=lambda m: _log(rank, m), store=store,
                        heartbeat_period_s=float(os.environ.get("DTF_HEARTBEAT_PERIOD_S", "5.0")),
                        fatal_grace_s=float(os.environ.get("DTF_FATAL_GRACE_S", "20")))
        sv.prepare_or_wait_for_session()
        sv.start_standard_services()

    import contextlib
    timer = None
    if FLAGS.profile:
        from ..utils import trace
        _log(rank, "profiling: roctx ranges %s" % ("on" if trace.enable(True) else "unavailable (libroctx64 not found)"))
        timer = trace.StepTimer()  # host-side split of the loop: launching steps / evaluation / summaries

    def phase(name):
        return timer.phase(name) if timer is not None else contextlib.nullcontext()
    start_step = engine.global_step
    flushed = start_step
    start_time = time.time()
    step = start_step
    rc = 0
    burst = 1
    try:
        while step < FLAGS.training_steps and not (sv is not None and sv.should_stop(steps=burst)):
            if FLAGS.fault_inject_step >= 0 and step == FLAGS.fault_inject_step and rank == world - 1:
                if os.environ.get("DTF_FAULT_KIND", "raise") == "die":
                    # the process vanishes (a killed worker, a lost node): nothing is said to anybody
                    print("rank %d: injected death at step %d" % (rank, step), flush=True)
                    os._exit(9)
                raise RuntimeError("injected fault at step %d" % step)
            # run up to the next point where the host has something to do (evaluation, summary flush, fault
            # injection, end of training; the stop flag is polled at least every 64 steps): the engine replays
            # several steps per graph launch in between
            interval = max(1, FLAGS.eval_step_interval)
            burst = min(FLAGS.training_steps - step, interval * -(-step // interval) + 1 - step,
                        512 - (step - flushed), 64)
            if FLAGS.fault_inject_step > step:
                burst = min(burst, FLAGS.fault_inject_step - step)
            burst = max(1, burst)
            if sv is not None:
                burst = sv.steps_allowed(step, burst)  # (an agreed stop step is not overshot)
            with phase("train steps (launch)"):
                trainer.train_steps(burst)
            if timer is not None:
                timer.add_images(burst * FLAGS.batch_size * world)
            if trainer.ipc is not None and trainer.comm == "ipc" and not trainer.ensure_healthy():
                # an in-kernel wait of the direct exchange expired on some rank: that step was skipped on every rank that
                # could know, the job has agreed on it, fallen back to the collective path and re-synchronised the replicas
                _log(rank, "gradient exchange fell back to %s at step %d" % (trainer.comm, engine.global_step))
            step = engine.global_step
            last = step - 1
            if step - flushed >= 512:
                with phase("step summaries (sync + read ring)"):
                    _flush_step_summaries(writer, engine, flushed, step)
                flushed = step
            if last % FLAGS.eval_step_interval == 0:
                with phase("step summaries (sync + read ring)"):
                    _flush_step_summaries(writer, engine, flushed, step)
                flushed = step
                with phase("evaluation"):
                    _tl, test_acc = _sharded_eval(trainer, test_dd.images, test_dd.labels)
                    if FLAGS.eval_train_set:
                        _trl, train_acc = _sharded_eval(trainer, train_dd.images, train_dd.labels)
                    else:
                        train_acc = float("nan")
                _log(rank, "Iter " + str(last) + ", test_acc: " + str(test_acc) + ", train_acc = " + str(train_acc))
                if FLAGS.check_replicas and not trainer.replicas_in_sync():
                    raise ConsistencyError("replica parameter checksums diverged at step %d" % last)
                if FLAGS.check_replicas and world > 1 and step < FLAGS.training_steps - 2:
                    # one checked step: what the optimizer consumed vs. the host-ordered sum of the local gradients
                    dev = trainer.check_exchange()
                    step = engine.global_step
                    _log(rank, "checked step %d: exchanged gradient deviates %.2g from the sum of the local gradients"
                         % (step - 1, dev))
                if writer is not None:
                    s = Summary().scalar("eval/test_accuracy", test_acc).scalar("eval/train_accuracy", train_acc)
                    if FLAGS.variable_summaries:
                        if hasattr(trainer, "sync_params"):
                            trainer.sync_params()
                        for name, v in engine.param_views().items():
                            s.merge(variable_summaries(name, v))
                    writer.add_summary(s, last)
        if hasattr(trainer, "materialize") and (sv is None or sv.clean_stop()):
            # a schedule with sharded optimizer state assembles the state of THIS step for the final checkpoint (collective:
            # every rank left the loop at the same step); after a failure the last agreed snapshot is saved instead
            trainer.materialize(snapshot=True)
    except BaseException as e:  # noqa: BLE001
        rc = 1
        if sv is not None:
            sv.request_stop(e)
        _log(0, "rank %d: training aborted: %r" % (rank, e))
        if not distributed or world == 1:
            raise
    finally:
        if hasattr(engine, "stream"):
            engine.stream.synchronize()
        elapsed = time.time() - start_time
        done = max(0, engine.global_step - start_step)
        _flush_step_summaries(writer, engine, flushed, engine.global_step)
        _log(rank, "Training time: %3.2fs" % elapsed)
        if timer is not None:
            _log(rank, "host phases: " + timer.summary())
        if done:
            _log(rank, "Throughput: %.1f images/sec (%d steps x global batch %d)" % (
                done * FLAGS.batch_size * world / max(1e-9, elapsed), done, FLAGS.batch_size * world))
        if sv is not None:
            # the chief saves whenever it is still alive — its loop ended through should_stop(), or its collective was
            # torn down under it by a peer that failed: parameters only change after a completed gradient exchange, so
            # its state is that of the last completed step (tf.train.Supervisor's chief saves on stop the same way).
            # Not after its OWN consistency / kernel error: Supervisor.stop() skips the save then (state_is_suspect).
            sv.stop(final_save=True)
        elif rank == 0:
            path = ckpt.Saver(engine, with_global_step=False).save(FLAGS.checkpoint_path)
            _log(rank, "Saved checkpoint " + path)
        if writer is not None:
            writer.close()
        if store is not None:
            store.close()
        if store_server is not None:
            store_server.stop()
        if server is not None:
            server.report_done()
    if sv is not None and engine.global_step < FLAGS.training_steps:
        _log(rank, "stopped at global step %d of %d (%s)" % (engine.global_step, FLAGS.training_steps,
                                                             sv.stop_reason or "stop requested"))
        return rc or 3  # (no barrier: the rank that failed is not coming)
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized() and rc == 0:
            dist.barrier()
            dist.destroy_process_group()
    return rc


def main(argv: Optional[List[str]] = None, distributed: bool = False) -> int:
    from .. import flags
    FLAGS, _unparsed = flags.parse_known(flags.build_mnist_parser(distributed), argv)
    return run(FLAGS, distributed)
