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
    width, height = float(im.size[0]), float(im.size[1])
    canvas = Image.new("L", (28, 28), 255)
    if width > height:
        nheight = int(round((20.0 / width * height), 0))
        if nheight == 0:
            nheight = 1
        img = im.resize((20, nheight), Image.LANCZOS).filter(ImageFilter.SHARPEN)
        wtop = int(round(((28 - nheight) / 2), 0))
        canvas.paste(img, (4, wtop))
    else:
        nwidth = int(round((20.0 / height * width), 0))
        if nwidth == 0:
            nwidth = 1
        img = im.resize((nwidth, 20), Image.LANCZOS).filter(ImageFilter.SHARPEN)
        wleft = int(round(((28 - nwidth) / 2), 0))
        canvas.paste(img, (wleft, 4))
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
