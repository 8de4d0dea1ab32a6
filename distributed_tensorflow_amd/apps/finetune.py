"""Inception-v3 *full* fine-tune driver (BASELINE config 5) — ``retrain*.py --finetune``.

The reference trains only the 2048->C head on frozen bottlenecks (``retrain1/retrain.py:372-476``); this mode keeps
the reference's data contract (image folders -> SHA-1 split ``:78-128``, class-balanced random sampling ``:322-354``,
optional distortions ``:137-165``, log line ``:455-457``, final test accuracy ``:467``, label file ``:474-475``) but
back-propagates through the whole network with the native training kernels, data-parallel over RCCL.

The decoded + resized training images live in HBM as one bf16 NHWC tensor (1.43 MB/image: 100k images fit in half
of one MI355X's 288 GB), so a step samples rows on the device instead of decoding JPEGs; with distortions enabled
the batch is re-decoded and augmented on the fly by the fused preprocessing kernel.
"""
from __future__ import annotations

import os
import random
import time
from datetime import datetime

import numpy as np


def _log(rank: int, msg: str) -> None:
    if rank == 0:
        print(msg, flush=True)


def _load_split(index, preproc, category: str, rank: int, world: int, shard: bool):
    """(images bf16 [n,299,299,8], labels int32 [n]) of one split; training is sharded rank-strided."""
    import torch
    rows = index.rows_of(category)
    if shard:
        rows = rows[rank::world]
    if len(rows) and preproc.device.type == "cuda" and os.environ.get("DTF_EXTRACT_PIPELINE", "1") != "0":
        # the HBM image cache is filled through the decode pipeline (decode processes -> pinned ring -> side-stream H2D +
        # resize), written batch by batch into one preallocated tensor
        from ..data.pipeline import DecodePipeline
        out = torch.empty(len(rows), 299, 299, 8, dtype=torch.bfloat16, device=preproc.device)
        pipe = DecodePipeline(preproc.device, batch_size=64)
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


def run(FLAGS, image_lists, image_dir: str, rank: int, world: int, device, seed: int) -> int:
    import torch

    from ..data import image_lists as il
    from ..data.bottleneck import Distortion, ImagePreprocessor
    from ..models.inception_train import make_inception_trainer
    from ..models.inception_v3 import InceptionV3
    from ..parallel.finetune_dp import FinetuneDataParallel
    from ..utils import checkpoint as ckpt
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
    dp = FinetuneDataParallel(trainer, bucket_mb=FLAGS.bucket_mb)
    preproc = ImagePreprocessor(device)
    distortion = Distortion(FLAGS.flip_left_right, FLAGS.random_crop, FLAGS.random_scale, FLAGS.random_brightness)
    rng = random.Random(seed * 7919 + rank)

    t0 = time.time()
    train_x, train_y = (None, None)
    if not distortion.enabled:
        train_x, train_y = _load_split(index, preproc, "training", rank, world, shard=True)
        if train_x is None:
            print("rank %d has no training images" % rank)
            return -1
        _log(rank, "HBM image cache: %d training images on this rank (%.1f MB) in %.2fs" % (
            train_x.shape[0], train_x.numel() * 2 / 1e6, time.time() - t0))
    train_rows = index.rows_of("training")
    # class-balanced sampling like get_random_cached_bottlenecks: random class, then random image of that class
    by_class = {}
    if train_y is not None:
        ty = train_y.cpu().numpy()
        for c in range(C):
            by_class[c] = np.nonzero(ty == c)[0]
    gen = torch.Generator(device="cpu").manual_seed(seed + 1000 * rank)

    def sample_batch():
        if distortion.enabled:
            paths, ys = [], []
            for _ in range(B):
                c = rng.randrange(C)
                label = list(image_lists.keys())[c]
                paths.append(il.get_image_path(image_lists, label, rng.randrange(il.MAX_NUM_IMAGES_PER_CLASS + 1), image_dir,
                                               "training"))
                ys.append(c)
            return preproc.load(paths, distortion, rng), torch.tensor(ys, dtype=torch.int32, device=device)
        cls = torch.randint(0, C, (B,), generator=gen).tolist()
        rows = []
        for c in cls:
            pool = by_class[c] if len(by_class[c]) else np.arange(train_x.shape[0])
            rows.append(int(pool[int(torch.randint(0, len(pool), (1,), generator=gen))]))
        r = torch.tensor(rows, device=device)
        return train_x.index_select(0, r), train_y.index_select(0, r)

    writer = FileWriter(os.path.join(FLAGS.summaries_dir, "finetune")) if rank == 0 else None
    steps = FLAGS.training_steps
    start = time.time()
    for i in range(steps):
        x, y = sample_batch()
        dp.train_step(x, y)
        if i % FLAGS.eval_step_interval == 0 or i + 1 == steps:
            loss, acc = dp.mean_metrics()
            if writer is not None:
                writer.add_summary(Summary().scalar("cross_entropy", loss).scalar("accuracy", acc), i)
            _log(rank, "%s: Step %d: Train accuracy = %.1f%% Cross entropy = %f" % (datetime.now(), i, acc * 100, loss))
    if device.type == "cuda":
        torch.cuda.synchronize()
    train_time = time.time() - start
    _log(rank, "Training time: %3.2fs" % train_time)
    if steps:
        _log(rank, "Fine-tune throughput: %.1f images/sec (global batch %d, %d GPU%s)" % (
            steps * B * world / max(train_time, 1e-9), B * world, world, "" if world == 1 else "s"))
    in_sync = dp.replicas_in_sync()
    _log(rank, "replicas in sync: %s" % in_sync)

    # ---- final test accuracy with the trained weights folded into the inference network (moving statistics)
    test_acc = float("nan")
    if rank == 0:
        test_x, test_y = _load_split(index, preproc, "testing", 0, 1, shard=False)
        if test_x is not None and native_ok:
            net = InceptionV3(device, seed=seed)
            trainer.export_to(net)
            W = trainer.param("final_training_ops/weights/final_weights")
            b = trainer.param("final_training_ops/biases/final_biases")
            correct = 0
            for j in range(0, test_x.shape[0], 64):
                feats = net.forward(test_x[j:j + 64].contiguous(), use_graph=False)
                correct += int(((feats @ W + b).argmax(1) == test_y[j:j + 64].long()).sum())
            test_acc = correct / test_x.shape[0]
            print("Final test accuracy = %.1f%% (N=%d)" % (test_acc * 100, test_x.shape[0]), flush=True)
        tensors = {k: trainer.layout.view(trainer.P, k).detach().cpu().numpy() for k in trainer.layout.entries}
        if hasattr(trainer, "moving_mean"):
            for name, s in trainer.graph.convs.items():
                o = trainer.stat_off[name]
                tensors[name + "/batchnorm/moving_mean"] = trainer.moving_mean[o:o + s.cout].cpu().numpy()
                tensors[name + "/batchnorm/moving_variance"] = trainer.moving_var[o:o + s.cout].cpu().numpy()
        tensors["_dtf/labels"] = np.frombuffer("\n".join(image_lists.keys()).encode(), np.uint8).copy()
        tensors["_dtf/finetuned"] = np.asarray(1, np.int64)
        ckpt.save_tensors(FLAGS.output_graph, tensors)
        with open(FLAGS.output_labels, "w") as f:
            f.write("\n".join(image_lists.keys()) + "\n")
        writer.close()
    _log(rank, "Total time: %3.2fs:" % (time.time() - begin))
    return 0 if in_sync else 3
