"""Inception-v3 transfer-learning driver shared by ``retrain1/retrain.py`` and
``retrain2/retrain2.py`` (component C37; reference ``main``: ``retrain1/retrain.py:372-476``,
``retrain2/retrain2.py:366-508``).

Flow: wipe + recreate ``summaries_dir`` -> (optional) model download -> build the backbone -> image
lists with the SHA-1 split -> bail out on 0/1 classes -> new softmax head -> batched bottleneck
extraction into the HBM cache (or the distortion pipeline) -> per step: train batch + validation
batch, SGD step, train/validation summaries -> every ``eval_step_interval`` steps the reference log
line -> final test accuracy on the whole test split -> export bundle + ``retrained_labels.txt``.

Design differences: rank 0 only touches the filesystem and extraction is sharded
over ranks + all-gathered (Q9); the per-step work never leaves the GPU (Q16); ``--eval_step_interval``
is honoured by both scripts (Q14); ``--print_misclassified_test_images`` does what its name says
(Q13); with no network the backbone is random-init (``--model_dir`` weights are loaded when a
converted ``inception_v3_weights.npz`` is present).
"""
from __future__ import annotations

import os
import random
import shutil
import sys
import time
from datetime import datetime
from typing import List, Optional

import numpy as np

DATA_URL = "http://download.tensorflow.org/models/image/imagenet/inception-2015-12-05.tgz"  # retrain1/retrain.py:27


def maybe_download_and_extract(model_dir: str, allow_network: bool = False) -> Optional[str]:
    """``maybe_download_and_extract`` (``retrain1/retrain.py:47-62``).  Kept optional: build/CI boxes have
    no network, in which case the backbone stays random-init.  Returns the extracted ``.pb`` path if present."""
    os.makedirs(model_dir, exist_ok=True)
    filename = DATA_URL.split("/")[-1]
    filepath = os.path.join(model_dir, filename)
    pb = os.path.join(model_dir, "classify_image_graph_def.pb")
    if os.path.exists(pb):
        return pb
    if not os.path.exists(filepath) and allow_network:
        import tarfile
        import urllib.request

        def _progress(count, block_size, total_size):
            sys.stdout.write("\r>> Downloading %s %.1f%%" % (filename, float(count * block_size) / float(total_size) * 100.0))
            sys.stdout.flush()
        try:
            filepath, _ = urllib.request.urlretrieve(DATA_URL, filepath, _progress)
            print("\nSuccessfully downloaded", filename, os.stat(filepath).st_size, "bytes.")
            tarfile.open(filepath, "r:gz").extractall(model_dir)
        except Exception as e:  # noqa: BLE001
            print("download failed (%r): using a random-init backbone" % (e,))
    return pb if os.path.exists(pb) else None


def _log(rank: int, msg: str) -> None:
    if rank == 0:
        print(msg, flush=True)


def _head_summary(head, prefix_scalars, batch=None):
    """What ``merged`` holds for the retrain graph (``retrain1/retrain.py:249-258,271-283,295``): the variable
    summaries of the new layer, the ``pre_activations`` / ``activations`` histograms of ``batch`` = (bottlenecks [B,2048])
    and the cross-entropy / accuracy scalars."""
    from ..utils.summary import Summary, variable_summaries
    s = Summary()
    s.merge(variable_summaries("final_training_ops/weights", head.weights))
    s.merge(variable_summaries("final_training_ops/biases", head.biases))
    if batch is not None:
        probs, logits = head.activations(batch)  # (the head's own forward kernel on a GPU)
        s.histogram("final_training_ops/Wx_plus_b/pre_activations", logits)
        s.histogram("final_training_ops/activations", probs)
    for k, v in prefix_scalars.items():
        s.scalar(k, v)
    return s


class _HeadSaver:
    """``tf.train.Saver`` over the variables the retrain2 Supervisor checkpoints (``retrain2/retrain2.py:418,426``):
    ``final_weights``, ``final_biases`` and the global step, under their TF names."""

    def __init__(self, head, max_to_keep: int = 5):
        self.head, self.max_to_keep, self._kept = head, max_to_keep, []

    def save(self, save_path: str, global_step=None) -> str:
        from ..utils import checkpoint as ckpt
        path = save_path if global_step is None else "%s-%d" % (save_path, global_step)
        ckpt.save_tensors(path, self.head.tf_tensors())
        ckpt.update_checkpoint_state(os.path.dirname(os.path.abspath(path)) or ".", path)
        self._kept.append(path)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            if old != path and os.path.exists(old):
                os.remove(old)
        return path

    def restore(self, path: str) -> int:
        from ..utils import checkpoint as ckpt
        self.head.load_tf_tensors(ckpt.load_tensors(path))
        return self.head.global_step


class _HeadTrainer:
    """The little of a trainer the Supervisor needs, over a ``RetrainHead``: rank/world, the stop-flag reduction and the
    chief's broadcast of parameters + global step."""

    def __init__(self, head, rank: int, world: int):
        self.engine, self.rank, self.world = head, rank, world

    def broadcast_state(self, src: int = 0) -> None:
        import torch
        import torch.distributed as dist
        from ..parallel.ddp import broadcast_
        broadcast_(self.engine.P, src=src)
        t = torch.tensor([self.engine.global_step], dtype=torch.int64,
                         device="cpu" if dist.get_backend() == "gloo" else self.engine.device)
        dist.broadcast(t, src=src)
        self.engine.global_step = int(t.item())

    def any_rank(self, flag: bool) -> bool:
        if self.world == 1:
            return bool(flag)
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32,
                         device="cpu" if dist.get_backend() == "gloo" else self.engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))


def run(FLAGS, distributed: bool) -> int:
    import torch

    from ..data import image_lists as il
    from ..data.bottleneck import (BottleneckCache, Distortion, ImagePreprocessor, get_random_distorted_bottlenecks)
    from ..models.retrain_head import RetrainHead, TinyBackbone
    from ..parallel import cluster as cl
    from ..parallel.ddp import init_process_group
    from ..utils import checkpoint as ckpt
    from ..utils.summary import FileWriter

    begin_time = time.time()
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
    seed = FLAGS.seed if FLAGS.seed is not None else 20151205
    rng = random.Random(seed * 7919 + rank)

    if rank == 0:  # Q9: only one process wipes the summaries dir
        if os.path.exists(FLAGS.summaries_dir):
            shutil.rmtree(FLAGS.summaries_dir)
        os.makedirs(FLAGS.summaries_dir, exist_ok=True)

    # ---- data: real image_dir, or a generated separable synthetic set (BASELINE config 3)
    image_dir = FLAGS.image_dir
    if FLAGS.synthetic or not os.path.exists(image_dir):
        if not FLAGS.synthetic and FLAGS.synthetic is not None:
            print("Image directory '" + image_dir + "' not found.")
            return -1
        image_dir = os.path.join(FLAGS.bottleneck_dir, "_synthetic_images")
        if rank == 0 and not os.path.exists(image_dir):
            il.make_synthetic_image_dir(image_dir, num_classes=FLAGS.synthetic_classes,
                                        images_per_class=FLAGS.synthetic_images_per_class, seed=seed)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        _log(rank, "using synthetic image set at " + image_dir)
    image_lists = il.create_image_lists(image_dir, FLAGS.testing_percentage, FLAGS.validation_percentage,
                                        verbose=(rank == 0))
    class_count = len(image_lists.keys())
    if class_count == 0:
        print("No valid folders of images found at " + image_dir)
        return -1
    if class_count == 1:
        print("Only one valid folder of images found at " + image_dir +
              " - multiple classes are needed for classification.")
        return -1
    if getattr(FLAGS, "finetune", False):
        from . import finetune
        rc = finetune.run(FLAGS, image_lists, image_dir, rank, world, device, seed, distributed=distributed, server=server)
        if server is not None:
            server.report_done()
        if world > 1:
            import torch.distributed as dist
            if rc == 0:
                dist.barrier()
            dist.destroy_process_group()
        return rc
    index = il.build_image_index(image_lists, image_dir)

    backbone_kind = getattr(FLAGS, "backbone", "auto")
    if backbone_kind == "auto":
        backbone_kind = "inception" if device.type == "cuda" else "tiny"
    # (the native bf16 forward needs the GPU; on a CPU-only box the Inception backbone runs as the fp32 reference forward)
    fp32_backbone = backbone_kind == "inception" and (getattr(FLAGS, "backbone_precision", "bf16") == "fp32"
                                                      or device.type != "cuda")
    # the decode processes of the extraction pipeline start now, so they are warm by the time the backbone is on the GPU
    pipeline = None
    if os.environ.get("DTF_EXTRACT_PIPELINE", "1") != "0" and (device.type == "cuda" or os.environ.get("DTF_EXTRACT_PIPELINE") == "1"):
        from ..data.pipeline import DecodePipeline
        # reference recipe by default: full-size decode, bilinear resize on the GPU (retrain1/retrain.py:34-35: DecodeJpeg
        # -> ResizeBilinear inside the graph); --decode_draft asks libjpeg for a reduced-scale decode of large sources
        pipeline = DecodePipeline(device, batch_size=FLAGS.extract_batch_size,
                                  draft=(2 * 299 if getattr(FLAGS, "decode_draft", False) else 0),
                                  fp32=fp32_backbone)

    # ---- backbone
    if backbone_kind == "inception":
        from ..models.inception_v3 import InceptionV3
        pb = maybe_download_and_extract(FLAGS.model_dir)
        net = InceptionV3(device, seed=seed)
        need_calibration = False
        npz = os.path.join(FLAGS.model_dir, "inception_v3_weights.npz")
        if pb is not None:
            # the reference's own artefact (retrain1/retrain.py:66-74), parsed without TensorFlow
            from ..utils.graphdef import inception_weights_from_graphdef
            layers = inception_weights_from_graphdef(pb)
            net.load_tf_weights(layers)
            _log(rank, "loaded %d Inception-v3 conv layers from %s" % (len(layers), pb))
        elif os.path.exists(npz):
            z = np.load(npz)
            layers = {}
            for k in z.files:
                layer, field = k.rsplit("__", 1)
                layers.setdefault(layer.replace("__", "/"), {})[field] = z[k]
            net.load_tf_weights(layers)
            _log(rank, "loaded Inception-v3 weights from " + npz)
        else:
            _log(rank, "Inception-v3 backbone: random-init weights (no pretrained artefact offline)")
            need_calibration = True
        backbone = net.forward
        if fp32_backbone:
            from ..models.inception_v3 import Fp32Backbone
            backbone = Fp32Backbone(net)
            if device.type == "cuda":
                _log(rank, "backbone precision: fp32 filters, activations and accumulation (the reference graph's precision; "
                           "4.5 k images/s at batch 128 after ~9 s of library algorithm search; the default bf16 path does 27 k)")
            else:
                _log(rank, "no GPU: the Inception-v3 backbone runs as the fp32 reference forward on the CPU (a few images/s)")
    else:
        backbone = TinyBackbone(device, seed=seed)
        _log(rank, "backbone: TinyBackbone (a fixed random projection of the down-sampled image, %s) — this exercises the "
                   "retrain PLUMBING only; accuracies say nothing about Inception-v3.  Use --backbone inception on a GPU."
             % ("requested" if getattr(FLAGS, "backbone", "auto") == "tiny" else "chosen because no GPU is present"))
    preproc = ImagePreprocessor(device, fp32=fp32_backbone)
    if backbone_kind == "inception" and need_calibration and index.num_images > 0:
        # random-init only: BatchNorm statistics from one batch of this dataset (same images on every rank)
        pick = np.random.RandomState(seed).permutation(index.num_images)[:min(32, index.num_images)]
        calib = preproc.load([index.paths[i] for i in sorted(pick)])
        if device.type == "cuda":
            from ..models.inception_train import calibrate_batchnorm
            calibrate_batchnorm(net, calib.to(torch.bfloat16))
        else:
            from ..models.inception_v3 import calibrate_batchnorm_reference
            calibrate_batchnorm_reference(net, calib)
        if fp32_backbone:
            backbone.invalidate()
        _log(rank, "calibrated BatchNorm statistics of the random-init backbone on %d images" % len(pick))

    head = RetrainHead(class_count, device, learning_rate=FLAGS.learning_rate, double_softmax=FLAGS.double_softmax,
                       seed=seed, final_tensor_name=FLAGS.final_tensor_name)
    head.world = world
    if world > 1:
        from ..parallel.ddp import broadcast_
        broadcast_(head.P, src=0)
        import torch.distributed as dist
        if device.type == "cuda" and dist.get_backend() == "nccl" and getattr(FLAGS, "comm", "auto") != "ipc":
            # the head's 41 KB gradient all-reduce goes through the framework's own communicator, enqueued on the step's
            # stream (reference transport: retrain2/retrain2.py:377,411-416); all ranks or none
            from ..parallel.rccl_comm import RcclCommunicator
            ok, comm = 1, None
            try:
                comm = RcclCommunicator.create(rank, world, device, key="rccl/head")
            except Exception as e:  # noqa: BLE001
                ok = 0
                _log(0, "rank %d: native RCCL communicator unavailable (%r): process-group all-reduce" % (rank, e))
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                head.comm = comm
                _log(rank, "head gradients: native RCCL communicator (%d ranks), enqueued on the step's stream" % world)
            elif comm is not None:
                comm.close(abort=True)

    distortion = Distortion(FLAGS.flip_left_right, FLAGS.random_crop, FLAGS.random_scale, FLAGS.random_brightness)
    cache = BottleneckCache(index, device)
    t_extract = time.time()
    try:
        created = cache.extract(backbone, preproc, batch_size=FLAGS.extract_batch_size, rank=rank, world=world,
                                txt_dir=FLAGS.bottleneck_dir if FLAGS.bottleneck_txt_cache else None,
                                progress=(print if rank == 0 else None), pipeline=pipeline)
    finally:
        if pipeline is not None and not distortion.enabled:
            pipeline.close()  # (a distorted run keeps the decode processes: its training batches stream through them)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.time() - t_extract
    _log(rank, "bottleneck extraction: %d images on this rank in %.2fs (%.1f images/sec/rank, HBM cache %.1f MB)" % (
        created, dt, created / max(dt, 1e-9), cache.values.numel() * 4 / 1e6))
    if pipeline is not None and cache.extract_stats:
        _log(rank, "extraction pipeline: " + DecodePipeline.describe(cache.extract_stats))

    train_writer = validation_writer = None
    if rank == 0:
        graph = {"model": "inception_v3->softmax", "classes": list(image_lists.keys()), "backbone": backbone_kind}
        train_writer = FileWriter(os.path.join(FLAGS.summaries_dir, "train"), graph)
        validation_writer = FileWriter(os.path.join(FLAGS.summaries_dir, "validation"))

    # retrain2 wraps the loop in a Supervisor (retrain2/retrain2.py:423-431,445,508): the chief restores the newest
    # ./logs/model.ckpt-<step> (head + global step) and everyone receives it, the chief autosaves every
    # --save_model_secs, the loop runs on the restored global step and polls should_stop(), sv.stop() saves at the end.
    sv = None
    store = store_server = None
    if distributed:
        from ..parallel.supervisor import Supervisor
        if world > 1:
            store = server.connect_store() if server is not None else None
            if store is None:
                store, store_server = cl.job_store(rank, world)
        sv = Supervisor(_HeadTrainer(head, rank, world), is_chief=(rank == 0), logdir=FLAGS.log_dir,
                        saver=_HeadSaver(head), save_model_secs=FLAGS.save_model_secs, summary_writer=train_writer,
                        log_fn=lambda m: _log(rank, m), store=store,
                        heartbeat_period_s=float(os.environ.get("DTF_HEARTBEAT_PERIOD_S", "5.0")),
                        fatal_grace_s=float(os.environ.get("DTF_FATAL_GRACE_S", "20")))
        sv.prepare_or_wait_for_session()
        sv.start_standard_services()

    # Per-step summaries (retrain1/retrain.py:440-446 runs `merged` on the training batch AND on a validation batch on
    # every step and writes both): cross-entropy and accuracy of every step accumulate in a device ring and reach the two
    # writers at logging points, each under its true step index; the variable summaries and the pre_activations /
    # activations histograms are written at the logging points.
    RING = 1024
    ring = torch.zeros(RING, 4, dtype=torch.float32, device=device)  # train ce-sum, train correct, val ce-sum, val correct
    head.use_metrics_ring(ring)
    flushed = head.global_step
    # --step_histograms: the rest of `merged` on every step too (variable_summaries of the new layer's weights and biases,
    # the pre_activations / activations histograms, for the training and for the validation batch): one statistics launch
    # per tensor into a device ring, read back with the scalars
    stats = cap_t = cap_v = None
    if getattr(FLAGS, "step_histograms", False):
        from ..utils.summary import StatsRing
        RING = 256
        stats = StatsRing([("final_training_ops/weights", True), ("final_training_ops/biases", True),
                           ("final_training_ops/Wx_plus_b/pre_activations", False), ("final_training_ops/activations", False),
                           ("final_training_ops/weights", True), ("final_training_ops/biases", True),
                           ("final_training_ops/Wx_plus_b/pre_activations", False), ("final_training_ops/activations", False)],
                          device, slots=RING)
        cap_t = tuple(torch.empty(FLAGS.train_batch_size, class_count, dtype=torch.float32, device=device) for _ in range(2))

    def flush_ring(upto: int) -> None:
        nonlocal flushed
        if upto <= flushed:
            return
        t = ring.clone()
        if world > 1:  # (collective: every rank flushes at the same steps)
            from ..parallel.ddp import all_reduce_
            all_reduce_(t)
            t /= world
        host = t.cpu().numpy()
        per_step = stats.flush(flushed, upto) if stats is not None and train_writer is not None else {}
        if train_writer is not None:
            from ..utils.summary import Summary
            for st in range(flushed, upto):
                r = host[st % RING]
                ts = Summary().scalar("cross_entropy", r[0] / FLAGS.train_batch_size).scalar("accuracy", r[1] / FLAGS.train_batch_size)
                vs = Summary().scalar("cross_entropy", r[2] / max(1, val_rows_n)).scalar("accuracy", r[3] / max(1, val_rows_n))
                if st in per_step:  # entries 0-3: the training run of `merged`, 4-7: the validation run
                    full = per_step[st]
                    nh = len(full.histograms) // 2
                    ts.scalars += full.scalars[:len(full.scalars) // 2]
                    vs.scalars += full.scalars[len(full.scalars) // 2:]
                    ts.histograms += full.histograms[:nh]
                    vs.histograms += full.histograms[nh:]
                train_writer.add_summary(ts, st)
                validation_writer.add_summary(vs, st)
        ring.zero_()
        flushed = upto

    # --validation_batch_size -1 = the whole split (retrain1/retrain.py:553-560): the same rows on every step, built once
    whole_val = cache.sample_rows("validation", -1, seed=seed + 977) if FLAGS.validation_batch_size < 0 else None
    val_rows_n = int(whole_val.numel()) if whole_val is not None else FLAGS.validation_batch_size

    start_time = time.time()
    steps = FLAGS.training_steps
    i = head.global_step if sv is not None else 0
    start_step = i
    rc = 0
    distorted = None
    if distortion.enabled and pipeline is not None and steps > i:
        # reference get_random_distorted_bottlenecks (retrain1/retrain.py:300-319) runs 2 x batch sess.run calls per step;
        # here the whole schedule is drawn up front and streamed: decode + H2D of later steps run under the current one
        from ..data.bottleneck import distorted_bottleneck_stream
        distorted = distorted_bottleneck_stream(index, FLAGS.train_batch_size, "training", backbone, pipeline, distortion,
                                                rng, steps - i)
    try:
        while i < steps and not (sv is not None and sv.should_stop()):
            if FLAGS.fault_inject_step >= 0 and i == FLAGS.fault_inject_step and rank == world - 1:
                raise RuntimeError("injected fault at step %d" % i)
            if distortion.enabled:
                if distorted is not None:
                    feats, labels = next(distorted)
                else:
                    feats, labels = get_random_distorted_bottlenecks(index, FLAGS.train_batch_size, "training", backbone,
                                                                     preproc, distortion, rng)
                head.train_step(feats, labels, None, ring_slot=i % RING, capture=cap_t)
                train_src = (feats, labels, None)
            else:
                rows = cache.sample_rows("training", FLAGS.train_batch_size, seed=seed + rank)
                head.train_step(cache.values, cache.labels, rows, ring_slot=i % RING, capture=cap_t)
                train_src = (cache.values, cache.labels, rows)
            if stats is not None:  # (the activations are those of the forward pass, the variables those after the update)
                stats.record(i, 0, head.weights)
                stats.record(i, 1, head.biases)
                stats.record(i, 2, cap_t[1])
                stats.record(i, 3, cap_t[0])
            # the validation batch of this step (every step, like the reference's second sess.run(merged, ...))
            vrows = whole_val if whole_val is not None else cache.sample_rows("validation", FLAGS.validation_batch_size,
                                                                                seed=seed + 977)
            if vrows.numel() > 0:
                if stats is not None and (cap_v is None or cap_v[0].shape[0] != vrows.numel()):
                    cap_v = tuple(torch.empty(int(vrows.numel()), class_count, dtype=torch.float32, device=device) for _ in range(2))
                head.accumulate_metrics(cache.values, cache.labels, vrows, ring_slot=i % RING, column=2, capture=cap_v)
                if stats is not None:
                    stats.record(i, 4, head.weights)
                    stats.record(i, 5, head.biases)
                    stats.record(i, 6, cap_v[1])
                    stats.record(i, 7, cap_v[0])
            is_last_step = (i + 1 == steps)
            log_now = (i % FLAGS.eval_step_interval) == 0 or is_last_step
            if log_now or (i + 1 - flushed) >= RING:
                flush_ring(i + 1)
            if log_now:
                # metrics of the training batch after the update (what the reference's extra sess.run reports)
                ce, train_acc = head.evaluate(*train_src)
                vce, val_acc = head.evaluate(cache.values, cache.labels, vrows) if vrows.numel() > 0 else (float("nan"),) * 2
                if world > 1:
                    import torch.distributed as dist
                    t = torch.tensor([ce, train_acc, vce, val_acc], dtype=torch.float64,
                                     device="cpu" if dist.get_backend() == "gloo" else device)
                    dist.all_reduce(t)
                    ce, train_acc, vce, val_acc = (t / world).tolist()
                if train_writer is not None and stats is None:  # (with --step_histograms every step already carries them)
                    tb = train_src[0] if train_src[2] is None else train_src[0].index_select(0, train_src[2].long())
                    train_writer.add_summary(_head_summary(head, {}, tb), i)
                    vb = cache.values.index_select(0, vrows.long()) if vrows.numel() > 0 else None
                    validation_writer.add_summary(_head_summary(head, {}, vb), i)
                if distributed:  # retrain2's own line (retrain2/retrain2.py:481-487): comma-separated, N = validation rows
                    _log(rank, "%s: Step %d: Train accuracy = %.1f%%,Cross entropy = %f,Validation accuracy = %.1f%% (N=%d)" % (
                        datetime.now(), i, train_acc * 100, ce, val_acc * 100, int(vrows.numel())))
                else:            # retrain1/retrain.py:455-457
                    _log(rank, "%s: Step %d: Train accuracy = %.1f%% Cross entropy = %f Validation accuracy = %.1f%%" % (
                        datetime.now(), i, train_acc * 100, ce, val_acc * 100))
            i += 1
    except BaseException as e:  # noqa: BLE001
        rc = 1
        if sv is not None:
            sv.request_stop(e)
        _log(0, "rank %d: retraining aborted: %r" % (rank, e))
        if sv is None or world == 1:
            raise
    finally:
        if distorted is not None:
            distorted.close()
        if pipeline is not None and distortion.enabled:
            if rank == 0 and pipeline.stats:
                print("distorted training stream: " + type(pipeline).describe(pipeline.stats), flush=True)
            pipeline.close()
        if sv is not None:
            sv.stop(final_save=True)
    if rc != 0 or (sv is not None and i < steps):
        # a coordinated stop (a peer failed, or stop was requested): the chief has saved; relaunching resumes from there
        _log(rank, "stopped at global step %d of %d (%s)" % (i, steps, (sv.stop_reason if sv is not None else None) or "exception"))
        if rank == 0 and train_writer is not None:
            train_writer.close()
            validation_writer.close()
        if head.comm is not None:
            head.comm.close(abort=True)  # (a peer is gone or going: do not wait for it)
        if store is not None:
            store.close()
        if store_server is not None:
            store_server.stop()
        if server is not None:
            server.report_done()
        return rc or 3
    if device.type == "cuda":
        torch.cuda.synchronize()
    train_time = time.time() - start_time

    test_rows = cache.sample_rows("testing", FLAGS.test_batch_size, seed=seed + 31337)
    if test_rows.numel() > 0:
        _ce, test_accuracy, probs, pred = head.evaluate(cache.values, cache.labels, test_rows, want_probs=True)
        _log(rank, "Final test accuracy = %.1f%% " % (test_accuracy * 100))
        if FLAGS.print_misclassified_test_images and rank == 0:
            print("=== MISCLASSIFIED TEST IMAGES ===")
            truth = cache.labels.index_select(0, test_rows.long()).cpu().tolist()
            labels = list(image_lists.keys())
            for f, p, t in zip(cache.filenames(test_rows), pred.cpu().tolist(), truth):
                if p != t:
                    print("%70s  %s" % (f, labels[p]))
    else:
        test_accuracy = float("nan")
        _log(rank, "Final test accuracy = n/a (empty testing split)")
    _log(rank, "Training time: %3.2fs" % train_time)
    if steps > start_step:
        _log(rank, "Head training throughput: %.1f bottleneck-images/sec (global batch %d)" % (
            (steps - start_step) * FLAGS.train_batch_size * world / max(train_time, 1e-9), FLAGS.train_batch_size * world))

    if rank == 0:
        tf_t = head.tf_tensors()
        if backbone_kind == "inception":
            # retrained_graph.pb is what the reference writes (retrain1/retrain.py:470-473: convert_variables_to_constants
            # over the imported Inception graph + the new layer): ONE serialized GraphDef with the 2015 graph's node names,
            # DecodeJpeg/contents ... pool_3/_reshape -> final_training_ops -> final_result.  test.py reads it back without
            # TensorFlow and reproduces the features the head was trained on, whatever the weights' origin (pretrained .pb,
            # .npz, or random-init + calibrated batch-norm statistics).
            from ..utils.graphdef import write_inception_graphdef
            n_nodes = write_inception_graphdef(FLAGS.output_graph, net.graph, net.params,
                                               tf_t["final_training_ops/weights/final_weights"],
                                               tf_t["final_training_ops/biases/final_biases"], FLAGS.final_tensor_name)
            _log(rank, "frozen graph: %s (%d nodes, %.1f MB)" % (FLAGS.output_graph, n_nodes,
                                                                  os.path.getsize(FLAGS.output_graph) / 1e6))
        else:
            # the CPU stand-in backbone has no TensorFlow graph to name: the native tensor container
            tensors = dict(tf_t)
            tensors["_dtf/final_tensor_name"] = np.frombuffer(FLAGS.final_tensor_name.encode(), np.uint8).copy()
            tensors["_dtf/backbone"] = np.frombuffer(backbone_kind.encode(), np.uint8).copy()
            tensors["_dtf/backbone_seed"] = np.asarray(seed, np.int64)
            tensors["backbone/tiny/proj"] = backbone.proj.detach().cpu().numpy()
            tensors["_dtf/labels"] = np.frombuffer("\n".join(image_lists.keys()).encode(), np.uint8).copy()
            ckpt.save_tensors(FLAGS.output_graph, tensors)
            # the frozen head as a GraphDef too (reference tensor names), for TensorFlow-side tooling
            from ..utils.graphdef import write_retrained_graph
            write_retrained_graph(os.path.splitext(FLAGS.output_graph)[0] + "_frozen_head.pb",
                                  tf_t["final_training_ops/weights/final_weights"],
                                  tf_t["final_training_ops/biases/final_biases"], FLAGS.final_tensor_name)
        with open(FLAGS.output_labels, "w") as f:
            f.write("\n".join(image_lists.keys()) + "\n")
        train_writer.close()
        validation_writer.close()
    _log(rank, "Total time: %3.2fs:" % (time.time() - begin_time))
    if head.comm is not None:
        head.comm.close()
    if store is not None:
        store.close()
    if store_server is not None:
        store_server.stop()
    if server is not None:
        server.report_done()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main(argv: Optional[List[str]] = None, distributed: bool = False) -> int:
    from .. import flags
    parser = flags.build_retrain_parser(distributed)
    parser.add_argument("--backbone", type=str, default="auto", help="auto | inception | tiny (CPU plumbing)")
    parser.add_argument("--backbone_precision", type=str, default="bf16", choices=["bf16", "fp32"],
                        help="bf16: the native MFMA forward (default).  fp32: filters, activations and accumulation in fp32 "
                             "like the reference's frozen graph, through the library convolutions (4.5 k instead of 27 k images/s)")
    FLAGS, _unparsed = flags.parse_known(parser, argv)
    return run(FLAGS, distributed)
