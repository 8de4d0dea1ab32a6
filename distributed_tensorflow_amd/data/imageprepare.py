"""MNIST inference preprocessor (component C38).

Same recipe as ``imageprepare`` in ``demo1/test.py:12-42``: grey-scale, scale the long
side to 20 px keeping aspect (min 1 px), antialiased resize + ``SHARPEN``, paste onto a
white 28x28 canvas (offset ``(4, top)`` or ``(left, 4)``), invert and normalise
``(255 - x) / 255`` so 0 = white background and 1 = black ink.

``Image.ANTIALIAS`` no longer exists in Pillow >= 10 (local Pillow is 12.x); it was an
alias of ``LANCZOS``, which is what is used here.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import numpy as np


def imageprepare(path: str) -> List[float]:
    from PIL import Image, ImageFilter

    im = Image.open(path).convert("L")
    size = (float(im.size[0]), float(im.size[1]))
    long_axis = 0 if size[0] > size[1] else 1          # 0: landscape, 1: portrait or square
    # the long side becomes 20 px, the short side keeps the aspect (at least 1 px) and is centred on the 28-px canvas
    short = max(1, int(round(20.0 / size[long_axis] * size[1 - long_axis], 0)))
    margin = int(round((28 - short) / 2, 0))
    new_size, corner = [0, 0], [0, 0]
    new_size[long_axis], new_size[1 - long_axis] = 20, short
    corner[long_axis], corner[1 - long_axis] = 4, margin
    digit = im.resize(tuple(new_size), Image.LANCZOS).filter(ImageFilter.SHARPEN)
    canvas = Image.new("L", (28, 28), 255)
    canvas.paste(digit, tuple(corner))
    px = np.asarray(canvas, dtype=np.float32).reshape(-1)
    return ((255.0 - px) * (1.0 / 255.0)).tolist()


def walk_images(root: str, exts: Tuple[str, ...] = (".jpg", ".jpeg", ".png", ".bmp")) -> List[str]:
    """Files under ``root`` in ``os.walk`` order (``demo1/test.py:7-8``)."""
    out: List[str] = []
    for r, _dirs, files in os.walk(root):
        for f in sorted(files):
            if f.lower().endswith(exts):
                out.append(os.path.join(r, f))
    return out


def prepare_batch(paths: Iterable[str]) -> np.ndarray:
    """All images at once -> float32 [N,784] (the reference rebuilds the graph per image, Q20)."""
    rows = [imageprepare(p) for p in paths]
    return np.asarray(rows, dtype=np.float32).reshape(len(rows), 784)
