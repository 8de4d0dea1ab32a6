"""Host statement of the device RNG (``csrc/hip/common.h``: Philox4x32-10) and of the dropout mask the native MNIST head
kernel draws from it (``csrc/hip/mnist_kernels.hip``: counter = (row*256 + thread, step, tag, 0), key = seed), so the
eager engine can train on exactly the native engine's dropout stream — what makes native-vs-eager trajectories comparable
step by step (``tests/test_f32_gpu.py``).  Reference semantics: ``tf.nn.dropout(h, keep_prob)``, ``demo1/train.py:108-111``.
"""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_TAG = 0x6D6E6973


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10: uint64 arrays holding 32-bit lanes in, 4 uint32-valued arrays out."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & _MASK, np.uint64(k1) & _MASK
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def dropout_keep_mask(batch: int, step: int, seed: int, keep_prob: float, hidden: int = 1024) -> np.ndarray:
    """``[batch, hidden]`` float32 of {0,1}: element (row, 4*t + q) is kept iff word q of
    Philox((row*256 + t, step, tag, 0), seed) >> 8, scaled to [0,1), is below ``keep_prob`` (fp32 compare, as on device)."""
    assert hidden == 1024
    rows = np.arange(batch, dtype=np.uint64)[:, None] * np.uint64(256) + np.arange(256, dtype=np.uint64)[None, :]
    z = np.zeros_like(rows)
    words = philox4x32_10(rows, z + np.uint64(step & 0xFFFFFFFF), z + np.uint64(_TAG), z, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack(words, axis=-1).reshape(batch, hidden)  # word q of thread t -> column 4*t + q
    unit = (u >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (unit < np.float32(keep_prob)).astype(np.float32)
