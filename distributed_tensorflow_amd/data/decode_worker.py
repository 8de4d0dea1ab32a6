"""JPEG decode workers of the extraction pipeline (``data/pipeline.py``).  Runs in worker PROCESSES: this module imports
nothing but numpy and PIL, so a spawned worker starts in a fraction of a second and never touches the GPU.

The reference decodes inside the TensorFlow graph, one image per ``sess.run`` (``DecodeJpeg`` fed from
``gfile.FastGFile(...).read()``, ``/root/reference/retrain1/retrain.py:168-180,228-231``).  Here a worker decodes a chunk of
a batch straight into a shared-memory slot that the parent has registered as pinned host memory; libjpeg's DCT-domain
down-scaling (PIL ``draft``) is requested whenever the source is at least twice the network's input size, which cuts the
decode time and the host->device bytes of large photographs by 4-64x with no effect on a 299 x 299 bilinear resample.
"""
from __future__ import annotations

from multiprocessing import shared_memory
from typing import Dict, List, Sequence, Tuple

import numpy as np

_SHM: Dict[str, shared_memory.SharedMemory] = {}


def _attach(name: str) -> shared_memory.SharedMemory:
    shm = _SHM.get(name)
    if shm is None:
        shm = shared_memory.SharedMemory(name=name)
        _SHM[name] = shm
    return shm


def _open(path: str, draft: int):
    from PIL import Image
    im = Image.open(path)
    if draft > 0:
        im.draft("RGB", (draft, draft))  # JPEG only: largest 1/2^k scale that keeps both sides >= draft
    return im


def probe(paths: Sequence[str], draft: int) -> List[Tuple[int, int]]:
    """(H, W) each image will decode to (header only)."""
    out = []
    for p in paths:
        try:
            with _open(p, draft) as im:
                out.append((im.size[1], im.size[0]))
        except Exception:
            out.append((0, 0))
    return out


def decode_into(buf, off: int, path: str, H: int, W: int, draft: int) -> None:
    with _open(path, draft) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        if im.size != (W, H):  # (the header lied, or a non-JPEG ignored the draft request)
            im = im.resize((W, H))
        a = np.asarray(im, dtype=np.uint8)
    np.frombuffer(buf, dtype=np.uint8, count=H * W * 3, offset=off)[:] = a.reshape(-1)


def decode_chunk(shm_name: str, items: Sequence[Tuple[str, int, int, int]], draft: int) -> Tuple[int, float]:
    """Decode ``(path, byte offset, H, W)`` items into the shared slot.  Returns (count, seconds spent)."""
    import time
    t0 = time.perf_counter()
    buf = _attach(shm_name).buf
    for path, off, H, W in items:
        decode_into(buf, off, path, H, W, draft)
    return len(items), time.perf_counter() - t0


def warm() -> int:
    from PIL import Image  # noqa: F401  (pay the import in the worker's start-up, not in its first chunk)
    return 0
