"""MNIST inference driver shared by ``demo1/test.py`` and ``demo2/test.py`` (components C38, C39).

Reference (``demo1/test.py:7-9,174-197``): for every file under ``imgs/`` the whole graph (incl.
Adam) is rebuilt, the checkpoint restored, the image shown with a blocking ``plt.show()``, and the
predicted digit printed.  Here the model is built once, all images are preprocessed
(``imageprepare``) and classified as one batch, display is opt-in and headless-safe, and the
demo2 flavour resolves the newest ``logs/model.ckpt-*`` instead of hard-coding step 3706 (Q20, Q21).
"""
from __future__ import annotations

import argparse
import os
from typing import List, Optional


def build_parser(default_ckpt: Optional[str], default_logdir: Optional[str]) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Classify digit images with a trained MNIST CNN checkpoint")
    p.add_argument("--imgs", type=str, default="imgs/")
    p.add_argument("--checkpoint", type=str, default=default_ckpt, help="checkpoint file (demo1: model/train.ckpt)")
    p.add_argument("--logdir", type=str, default=default_logdir, help="resolve the latest model.ckpt-* here (demo2)")
    p.add_argument("--device", type=str, default="auto")
    p.add_argument("--engine", type=str, default="auto")
    p.add_argument("--show", action="store_true", help="display each image with matplotlib (blocks like the reference)")
    return p


def run(args) -> List[int]:
    import torch

    from ..data.imageprepare import prepare_batch, walk_images
    from ..engine.mnist_engine import make_engine
    from ..utils import checkpoint as ckpt

    path = args.checkpoint
    if (path is None or not os.path.exists(path)) and args.logdir:
        path = ckpt.latest_checkpoint(args.logdir)
    if path is None or not os.path.exists(path):
        raise FileNotFoundError("no checkpoint found (checkpoint=%r, logdir=%r)" % (args.checkpoint, args.logdir))
    use_cuda = torch.cuda.is_available() and args.device != "cpu"
    kind = args.engine if args.engine != "auto" else ("native" if use_cuda else "eager")
    device = "cuda" if use_cuda else "cpu"
    files = walk_images(args.imgs)
    if not files:
        raise FileNotFoundError("no images under %r" % args.imgs)
    engine = make_engine(kind, None, batch_size=max(2, len(files)), device=device, eval_chunk=max(64, len(files)))
    step = ckpt.Saver(engine).restore(path)
    print("restored %s (step %d)" % (path, step))
    x = torch.from_numpy(prepare_batch(files)).to(device)
    probs, pred = engine.predict(x)
    pred = pred.cpu().tolist()
    for f, p in zip(files, pred):
        if args.show:
            try:
                import matplotlib
                if not os.environ.get("DISPLAY"):
                    matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                from PIL import Image
                plt.imshow(Image.open(f))
                plt.axis("off")
                plt.show()
            except Exception as e:  # headless box: keep going
                print("(display skipped: %r)" % (e,))
        print(os.path.basename(f))
        print("识别出该图片是数字:" + str(p))
    return pred


def main(argv=None, default_ckpt="model/train.ckpt", default_logdir=None):
    args, _ = build_parser(default_ckpt, default_logdir).parse_known_args(argv)
    return run(args)
