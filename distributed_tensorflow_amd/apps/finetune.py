"""Inception-v3 *full* fine-tune driver (BASELINE config 5) — ``retrain*.py --finetune``.

The reference trains only the 2048->C head on frozen bottlenecks (``retrain1/retrain.py:372-476``); this mode keeps
the reference's data contract (image folders -> SHA-1 split ``:78-128``, class-balanced random sampling ``:322-354``,
optional distortions ``:137-165``, log line ``:455-457`` / ``retrain2/retrain2.py:481-487``, final test accuracy ``:467``,
``retrained_graph.pb`` + label file ``:470-475``) but back-propagates through the whole network with the native training
kernels, data-parallel over RCCL.  ``retrain2.py --finetune`` runs under the same Supervisor as every distributed script
of the reference (``retrain2/retrain2.py:423-431,445,508``): the chief restores the newest ``./logs/model.ckpt-<step>``
(parameters, optimizer slots, moving statistics, step), everyone receives it, autosave every ``--save_model_secs``, stop
flag and heartbeats out of band, ``--fault_inject_step``.

The decoded + resized training images live in HBM as one bf16 NHWC tensor (1.43 MB/image: 100k images fit in half of one
MI355X's 288 GB); a step draws its batch ON THE DEVICE — the class-balanced sampler kernel of the retrain head
(``dtf_sample_rows``) and one gather kernel into a static input buffer — so the host issues launches only.  With
distortions enabled the batch is re-decoded and augmented on the fly by the fused preprocessing kernel.
"""
from __future__ import annotations

import os
import random
import time
from datetime import datetime
from typing import Optional

import numpy as np


def _log(rank: int, msg: str) -> None:
    if rank == 0:
        print(msg, flush=True)


def _load_split(index, preproc, category: str, rank: int, world: int, shard: bool):
    """(images bf16 [n,299,299,8], labels int32 [n]) of one split, label-major; training is sharded rank-strided."""
    import torch
    rows = index.rows_of(category)
    if shard:
        rows = rows[rank::world]
    if len(rows) and preproc.device.type == "cuda" and os.environ.get("DTF_EXTRACT_PIPELINE", "1") != "0":
        # the HBM image cache is filled through the decode pipeline (decode processes -> pinned ring -> side-stream H2D +
        # resize), written batch by batch into one preallocated tensor
        from ..data.pipeline import DecodePipeline
        out = torch.empty(len(rows), 299, 299, 8, dtype=torch.bfloat16, device=preproc.device)
        pipe = DecodePipeline(preproc.device, batch_size=64, draft=0)
        try:
            for first, count, x in pipe.batches([index.paths[r] for r in rows]):
                out[first:first + count].copy_(x[:count])
        finally:
            pipe.close()
        return out, torch.as_tensor(index.class_id[rows], dtype=torch.int32).to(preproc.device)
    xs, ys = [], []
    for i in range(0, len(rows), 32):
        chunk = rows[i:i + 32]
        xs.append(preproc.load([index.paths[r] for r in chunk]))
        ys.append(torch.as_tensor(index.class_id[chunk], dtype=torch.int32))
    if not xs:
        return None, None
    return torch.cat(xs), torch.cat(ys).to(xs[0].device)


class DeviceBatchSampler:
    """``get_random_cached_bottlenecks`` (``retrain1/retrain.py:322-340``: random class, then random image of that class,
    with replacement) over an HBM-resident image split, entirely on the device: ``dtf_sample_rows`` draws the rows
    (counter-based RNG keyed by (seed, draw)), ``dtf_gather_rows`` copies the picked images and labels into the caller's
    static batch buffers.  No per-step host list, no Python loop, no host->device copy; the draw counter itself lives in
    device memory, so the sampler can be part of a captured step graph.  CPU: the same draw with torch ops."""

    def __init__(self, x, y, num_classes: int, seed: int):
        import torch
        self.x, self.device, self.C, self.seed, self.draws = x, x.device, num_classes, int(seed), 0
        ys = y.cpu().numpy()
        if np.any(ys[1:] < ys[:-1]):  # (splits are loaded label-major; anything else is sorted once)
            order = np.argsort(ys, kind="stable")
            self.x = x = x[torch.as_tensor(order, device=x.device)]
            ys = ys[order]
        self.y = torch.as_tensor(ys, dtype=torch.int32, device=self.device)
        counts = np.bincount(ys, minlength=num_classes).astype(np.int64)
        self.counts = torch.as_tensor(counts, device=self.device)
        self.offsets = torch.as_tensor(np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.int64), device=self.device)
        self.row_bytes = x[0].numel() * x.element_size()
        self._rows = {}
        self._draw_t = torch.zeros(1, dtype=torch.int64, device=self.device) if self.device.type == "cuda" else None

    def sample_into(self, out_x, out_y) -> None:
        import torch
        B = int(out_x.shape[0])
        self.draws += 1
        if self.device.type == "cuda":
            from .. import native
            lib = native.hip()
            if B not in self._rows:
                self._rows[B] = torch.empty(B, dtype=torch.int32, device=self.device)
            rows = self._rows[B]
            st = torch.cuda.current_stream(self.device).cuda_stream
            native.check(lib.dtf_sample_rows_dev(self.offsets.data_ptr(), self.counts.data_ptr(), self.C, B, self.seed,
                                                 self._draw_t.data_ptr(), rows.data_ptr(), st), "sample_rows")
            native.check(lib.dtf_gather_rows(self.x.data_ptr(), rows.data_ptr(), out_x.data_ptr(), self.row_bytes, B,
                                             self.y.data_ptr(), out_y.data_ptr(), st), "gather_rows")
            return
        g = torch.Generator().manual_seed((self.seed * 1000003 + self.draws) % (2 ** 63))
        valid = torch.nonzero(self.counts > 0).reshape(-1)
        cls = valid[torch.randint(0, len(valid), (B,), generator=g)]
        r = torch.randint(0, 1 << 27, (B,), generator=g)
        rows = self.offsets[cls] + r % self.counts[cls]
        out_x.copy_(self.x.index_select(0, rows))
        out_y.copy_(self.y.index_select(0, rows))


def pair_allowed(at: int, *, start_step: int, warm_steps: int, steps: int, eval_step_interval: int, fault_inject_step: int) -> bool:
    """May steps ``at`` and ``at + 1`` run as one graph replay?  Not during the eager warm-up (autotuning, lazy allocations),
    not when either is a logging point (its metrics are read right after it; the pair's second step clears that slot), not
    across the injected fault, and not the last step (logged as well)."""
    if at - start_step < max(2, warm_steps) or at + 2 > steps - 1:
        return False
    if at % eval_step_interval == 0 or (at + 1) % eval_step_interval == 0:
        return False
    return fault_inject_step not in (at, at + 1)


class StepPairGraph:
    """Two consecutive training steps — batch sampling, forward, loss, backward, gradient exchange, optimizer — as ONE
    hipGraph replay.  Two, because the step alternates between two sets of batch-norm statistic
    accumulators and two metric slots (``inception_train.py``): after a pair everything is back where it started, so the
    captured pair replays unchanged.  One graph per starting parity, captured on first use after the eager warm-up steps
    (autotuning, lazy allocations).  Capture errors disable the graph for the rest of the run; the step then stays eager.

    ``lock`` is held while a capture is open: the Supervisor's saver thread synchronises the device in ``state_dict()``,
    which is not allowed while a stream of the process is capturing."""

    def __init__(self, step_fn, parity_fn, device, lock, log=lambda m: None):
        self.step_fn, self.parity_fn, self.device, self.lock, self.log = step_fn, parity_fn, device, lock, log
        self.graphs = {}
        self.enabled = True
        self.replays = 0

    def run(self) -> bool:
        """Run two steps through the graph; False (and nothing enqueued) when the graph is not available."""
        import torch
        if not self.enabled:
            return False
        par = self.parity_fn()
        g = self.graphs.get(par)
        if g is None:
            with self.lock:
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self.step_fn()
                        self.step_fn()
                except Exception as e:  # noqa: BLE001
                    self.enabled = False
                    self.log("step graph: capture failed (%r); the step stays eager" % (e,))
                    torch.cuda.synchronize(self.device)
                    return False
            if self.parity_fn() != par:
                self.enabled = False
                self.log("step graph: parity did not return to its start after two steps; the step stays eager")
                return False
            self.graphs[par] = g
        g.replay()
        self.replays += 1
        return True


class FinetuneEngine:
    """What the Supervisor and its saver see of the fine-tune: a global step and a state to snapshot / restore."""

    def __init__(self, trainer):
        import threading
        self.trainer = trainer
        self.global_step = 0
        self.capture_lock = threading.RLock()

    def state_dict(self):
        import torch
        with self.capture_lock:  # (never while the training thread has a graph capture open)
            if self.trainer.P.is_cuda:
                torch.cuda.synchronize(self.trainer.P.device)
            return {k: v.detach().cpu().numpy().copy() for k, v in self.trainer.state_tensors().items()}

    def load_state_dict(self, sd) -> None:
        import torch
        for k, t in self.trainer.state_tensors().items():
            if k not in sd:
                raise KeyError("checkpoint holds no %r (written with another optimizer?)" % k)
            src = torch.as_tensor(np.asarray(sd[k])).reshape(t.shape)
            if src.numel() != t.numel():
                raise ValueError("checkpoint %s has %d elements, the model %d (another class count?)" % (k, src.numel(), t.numel()))
            t.copy_(src.to(t.device, t.dtype))
        self.global_step = int(np.asarray(sd["step"]).reshape(-1)[0])


class FinetuneSaver:
    """``tf.train.Saver`` of the fine-tune job: ``logdir/model.ckpt-<global_step>`` in the native CRC container."""

    def __init__(self, engine: FinetuneEngine, max_to_keep: int = 3):
        self.engine, self.max_to_keep, self._kept = engine, max_to_keep, []

    def save(self, save_path: str, global_step=None) -> str:
        from ..utils import checkpoint as ckpt
        path = save_path if global_step is None else "%s-%d" % (save_path, global_step)
        sd = self.engine.state_dict()
        sd["step"] = np.asarray([self.engine.global_step], np.int32)  # (the host's count: SGD keeps no device step)
        ckpt.save_tensors(path, {"finetune/" + k: v for k, v in sd.items()})
        ckpt.update_checkpoint_state(os.path.dirname(os.path.abspath(path)) or ".", path)
        self._kept.append(path)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            if old != path and os.path.exists(old):
                os.remove(old)
        return path

    def restore(self, path: str) -> int:
        from ..utils import checkpoint as ckpt
        t = ckpt.load_tensors(path)
        self.engine.load_state_dict({k[len("finetune/"):]: v for k, v in t.items() if k.startswith("finetune/")})
        return self.engine.global_step


class _SvTrainer:
    """The little of a trainer the Supervisor needs."""

    def __init__(self, engine: FinetuneEngine, dp, rank: int, world: int):
        self.engine, self.dp, self.rank, self.world = engine, dp, rank, world

    def broadcast_state(self, src: int = 0) -> None:
        import torch
        import torch.distributed as dist
        self.dp.broadcast_state(src)
        t = torch.tensor([self.engine.global_step], dtype=torch.int64)
        if dist.get_backend() != "gloo":
            t = t.to(self.engine.trainer.P.device)
        dist.broadcast(t, src=src)
        self.engine.global_step = int(t.item())

    def any_rank(self, flag: bool) -> bool:
        if self.world == 1:
            return bool(flag)
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if dist.get_backend() != "gloo":
            t = t.to(self.engine.trainer.P.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    def synchronize(self) -> None:
        import torch
        if self.engine.trainer.P.is_cuda:
            torch.cuda.synchronize(self.engine.trainer.P.device)


def export_graph(trainer, image_lists, FLAGS, seed: int):
    """``retrained_graph.pb`` + ``retrained_labels.txt`` exactly as the head-only retrain writes them (one GraphDef of
    Inception-v3 + the new layer, ``retrain1/retrain.py:470-475``), so ``retrain{1,2}/test.py`` loads a fine-tuned network
    like any other.  Returns the inference network holding the trained weights and moving statistics."""
    from ..models.inception_v3 import InceptionV3
    from ..utils.graphdef import write_inception_graphdef
    net = InceptionV3(device=trainer.P.device if trainer.P.is_cuda else "cpu", seed=seed)
    trainer.export_to(net)
    W = trainer.param("final_training_ops/weights/final_weights").detach().cpu().numpy()
    b = trainer.param("final_training_ops/biases/final_biases").detach().cpu().numpy()
    write_inception_graphdef(FLAGS.output_graph, net.graph, net.params, W, b, FLAGS.final_tensor_name)
    with open(FLAGS.output_labels, "w") as f:
        f.write("\n".join(image_lists.keys()) + "\n")
    return net


def run(FLAGS, image_lists, image_dir: str, rank: int, world: int, device, seed: int, distributed: bool = False,
        server=None) -> int:
    import torch

    from ..data import image_lists as il
    from ..data.bottleneck import Distortion, ImagePreprocessor
    from ..models.inception_train import make_inception_trainer
    from ..models.inception_v3 import InceptionV3
    from ..models.retrain_head import RetrainHead
    from ..parallel.finetune_dp import FinetuneDataParallel
    from ..utils.summary import FileWriter, Summary

    begin = time.time()
    index = il.build_image_index(image_lists, image_dir)
    C = len(image_lists)
    B = FLAGS.finetune_batch_size
    native_ok = device.type == "cuda"
    init = InceptionV3(device="cpu", seed=seed)
    pb = os.path.join(FLAGS.model_dir, "classify_image_graph_def.pb")
    if os.path.exists(pb):
        from ..utils.graphdef import inception_weights_from_graphdef
        init.load_tf_weights(inception_weights_from_graphdef(pb))
        _log(rank, "fine-tuning from " + pb)
    else:
        _log(rank, "fine-tuning from random-init weights (no pretrained artefact offline)")
    trainer = make_inception_trainer("native" if native_ok else "eager", C, B, device, lr=FLAGS.learning_rate,
                                     optimizer=FLAGS.finetune_optimizer, seed=seed, init=init,
                                     double_softmax=FLAGS.double_softmax)
    dp = FinetuneDataParallel(trainer, bucket_mb=FLAGS.bucket_mb, comm=getattr(FLAGS, "comm", "auto"),
                              grad_dtype=getattr(FLAGS, "finetune_grad_dtype", "fp32"))
    engine = FinetuneEngine(trainer)
    preproc = ImagePreprocessor(device)
    distortion = Distortion(FLAGS.flip_left_right, FLAGS.random_crop, FLAGS.random_scale, FLAGS.random_brightness)
    rng = random.Random(seed * 7919 + rank)

    # ---- Supervisor (distributed script only, like the reference): restore + broadcast, autosave, stop protocol
    sv = store = store_server = None
    writer = FileWriter(os.path.join(FLAGS.summaries_dir, "finetune")) if rank == 0 else None
    if distributed:
        from ..parallel import cluster as cl
        from ..parallel.supervisor import Supervisor
        if world > 1:
            store = server.connect_store() if server is not None else None
            if store is None:
                store, store_server = cl.job_store(rank, world)
        sv = Supervisor(_SvTrainer(engine, dp, rank, world), is_chief=(rank == 0), logdir=FLAGS.log_dir,
                        saver=FinetuneSaver(engine), save_model_secs=FLAGS.save_model_secs, summary_writer=writer,
                        log_fn=lambda m: _log(rank, m), store=store,
                        heartbeat_period_s=float(os.environ.get("DTF_HEARTBEAT_PERIOD_S", "5.0")),
                        fatal_grace_s=float(os.environ.get("DTF_FATAL_GRACE_S", "20")), stop_margin_steps=4)
        sv.prepare_or_wait_for_session()
        sv.start_standard_services()

    # ---- data: the training shard and the validation split live in HBM
    t0 = time.time()
    sampler = val_sampler = None
    if not distortion.enabled:
        train_x, train_y = _load_split(index, preproc, "training", rank, world, shard=True)
        if train_x is None:
            print("rank %d has no training images" % rank)
            return -1
        sampler = DeviceBatchSampler(train_x, train_y, C, seed + 1000 * rank)
        _log(rank, "HBM image cache: %d training images on this rank (%.1f MB) in %.2fs" % (
            train_x.shape[0], train_x.numel() * 2 / 1e6, time.time() - t0))
    val_x, val_y = _load_split(index, preproc, "validation", rank, world, shard=False)
    if val_x is not None and val_x.shape[0] > 0:
        val_sampler = DeviceBatchSampler(val_x, val_y, C, seed + 977)
    labels_list = list(image_lists.keys())
    xb = torch.empty(B, 299, 299, 8, dtype=torch.bfloat16 if native_ok else torch.float32, device=device)
    yb = torch.empty(B, dtype=torch.int32, device=device)
    if sampler is not None and sampler.x.dtype != xb.dtype:
        xb = xb.to(sampler.x.dtype)
    val_metrics = torch.zeros(2, dtype=torch.float32, device=device)

    def next_batch():
        if sampler is not None:
            sampler.sample_into(xb, yb)
            return xb, yb
        paths, ys = [], []
        for _ in range(B):
            c = rng.randrange(C)
            paths.append(il.get_image_path(image_lists, labels_list[c], rng.randrange(il.MAX_NUM_IMAGES_PER_CLASS + 1),
                                           image_dir, "training"))
            ys.append(c)
        return preproc.load(paths, distortion, rng), torch.tensor(ys, dtype=torch.int32, device=device)

    def validation_accuracy() -> float:
        """Two class-balanced validation batches (forward only, batch statistics), averaged over the ranks."""
        if val_sampler is None:
            return float("nan")
        val_metrics.zero_()
        for _ in range(2):
            val_sampler.sample_into(xb, yb)
            trainer.evaluate_batch(xb, yb, val_metrics)
        return dp.mean_scalar(float(val_metrics[1].item()) / (2 * B))

    steps = FLAGS.training_steps
    i = engine.global_step
    start_step = i
    rc = 0
    start = time.time()
    warm_steps, t_warm = min(5, max(1, (steps - i) // 2)), None

    def one_step():
        x, y = next_batch()
        dp.train_step(x, y)

    # pairs of steps as one hipGraph replay wherever the host has nothing to do in between (no logging point, no fault
    # injection, no agreed stop step inside the pair); everything else, and the warm-up, runs eagerly
    pair = None
    if native_ok and sampler is not None and getattr(FLAGS, "finetune_graph", False) and dp.graph_capable():
        pair = StepPairGraph(one_step, lambda: trainer._fwd_parity, device, engine.capture_lock, log=lambda m: _log(rank, m))

    def pair_ok(at: int) -> bool:
        if pair is None or not pair.enabled:
            return False
        if not pair_allowed(at, start_step=start_step, warm_steps=warm_steps, steps=steps,
                            eval_step_interval=FLAGS.eval_step_interval, fault_inject_step=FLAGS.fault_inject_step):
            return False
        return sv is None or sv.steps_allowed(at, 2) >= 2  # (an agreed stop step is not overshot)

    try:
        while i < steps and not (sv is not None and sv.should_stop()):
            if FLAGS.fault_inject_step >= 0 and i == FLAGS.fault_inject_step and rank == world - 1:
                raise RuntimeError("injected fault at step %d" % i)
            if pair_ok(i) and pair.run():
                i += 2
                engine.global_step = i
                if hasattr(trainer, "_steps_host"):
                    trainer._steps_host += 2
                continue
            one_step()
            i += 1
            engine.global_step = i
            st = i - 1
            if i - start_step == warm_steps and device.type == "cuda":
                torch.cuda.synchronize()  # (steady-state clock: the first steps hold lazy initialisation and the first log)
                t_warm = time.time()
            if st % FLAGS.eval_step_interval == 0 or i == steps:
                loss, acc = dp.mean_metrics()
                vacc = validation_accuracy()
                if writer is not None:
                    writer.add_summary(Summary().scalar("cross_entropy", loss).scalar("accuracy", acc)
                                       .scalar("validation_accuracy", vacc), st)
                if distributed:
                    _log(rank, "%s: Step %d: Train accuracy = %.1f%%,Cross entropy = %f,Validation accuracy = %.1f%% (N=%d)" % (
                        datetime.now(), st, acc * 100, loss, vacc * 100, 2 * B * world))
                else:
                    _log(rank, "%s: Step %d: Train accuracy = %.1f%% Cross entropy = %f Validation accuracy = %.1f%%" % (
                        datetime.now(), st, acc * 100, loss, vacc * 100))
                err = dp.async_error()
                if err:
                    raise RuntimeError("RCCL reported an asynchronous error: %s" % err)
    except BaseException as e:  # noqa: BLE001
        rc = 1
        if sv is not None:
            sv.request_stop(e)
        _log(0, "rank %d: fine-tuning aborted: %r" % (rank, e))
        if sv is None or world == 1:
            raise
    finally:
        if sv is not None:
            sv.stop(final_save=True)

    def shutdown():
        if writer is not None:
            writer.close()
        if store is not None:
            store.close()
        if store_server is not None:
            store_server.stop()
        dp.close()

    if rc != 0 or (sv is not None and i < steps):
        _log(rank, "stopped at global step %d of %d (%s)" % (i, steps, (sv.stop_reason if sv is not None else None) or "exception"))
        shutdown()
        return rc or 3
    if device.type == "cuda":
        torch.cuda.synchronize()
    train_time = time.time() - start
    _log(rank, "Training time: %3.2fs" % train_time)
    if i > start_step:
        _log(rank, "Fine-tune throughput: %.1f images/sec (global batch %d, %d GPU%s; batch sampling %s)" % (
            (i - start_step) * B * world / max(train_time, 1e-9), B * world, world, "" if world == 1 else "s",
            "on the device" if sampler is not None else "decoded and distorted per step"))
        if t_warm is not None and i - start_step > warm_steps:
            t_end = start + train_time
            _log(rank, "Fine-tune steady state: %.1f images/sec (%d steps after %d warm-up steps, sampling, validation and "
                       "logging included; last train loss %.4f)" % (
                (i - start_step - warm_steps) * B * world / max(t_end - t_warm, 1e-9), i - start_step - warm_steps,
                warm_steps, dp.mean_metrics()[0]))
    if pair is not None:
        _log(rank, "step graph: %d of %d steps ran as two-step hipGraph replays (%d graph%s captured)" % (
            2 * pair.replays, i - start_step, len(pair.graphs), "" if len(pair.graphs) == 1 else "s"))
    in_sync = dp.replicas_in_sync()
    _log(rank, "replicas in sync: %s" % in_sync)

    # ---- export + final test accuracy with the trained weights folded into the inference network (moving statistics)
    if rank == 0:
        k_cal = int(getattr(FLAGS, "finetune_bn_recalibrate", 8))
        if k_cal > 0 and sampler is not None and hasattr(trainer, "bn_momentum"):
            # The moving statistics follow the batch statistics with TensorFlow's decay of 0.9997: after a short run they
            # are still the initial ones and the frozen network would not be the network that was trained.  Re-estimate
            # them as the plain average of the batch statistics over k training batches (momentum (j-1)/j at batch j).
            keep = trainer.bn_momentum
            for j in range(1, k_cal + 1):
                sampler.sample_into(xb, yb)
                trainer.bn_momentum = (j - 1.0) / j
                trainer.forward(xb, training=True)
            if k_cal % 2:  # (an even number of forwards keeps the statistic accumulators' alternation, see evaluate_batch)
                trainer.forward(xb, training=False)
            trainer.bn_momentum = keep
            _log(rank, "batch-norm moving statistics re-estimated over %d training batches for the export" % k_cal)
        net = export_graph(trainer, image_lists, FLAGS, seed)
        test_x, test_y = _load_split(index, preproc, "testing", 0, 1, shard=False)
        if test_x is not None and native_ok:
            head = RetrainHead(C, device)
            head.load_tf_tensors({"final_training_ops/weights/final_weights":
                                  trainer.param("final_training_ops/weights/final_weights").detach().cpu().numpy(),
                                  "final_training_ops/biases/final_biases":
                                  trainer.param("final_training_ops/biases/final_biases").detach().cpu().numpy()})
            correct = 0
            for j in range(0, test_x.shape[0], 64):
                probs = head.predict(net.forward(test_x[j:j + 64].contiguous(), use_graph=False))  # (the head's own kernel)
                correct += int((probs.argmax(1) == test_y[j:j + 64].long()).sum())
            print("Final test accuracy = %.1f%% (N=%d)" % (100.0 * correct / test_x.shape[0], test_x.shape[0]), flush=True)
    _log(rank, "Total time: %3.2fs:" % (time.time() - begin))
    shutdown()
    return 0 if in_sync else 3
