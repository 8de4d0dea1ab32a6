"""MNIST dataset reader + mini-batch sampler (components C1, C2).

Reference behaviour: ``input_data.read_data_sets("MNIST_data", one_hot=True)``
(``demo1/train.py:3,6``) reads four IDX ``.gz`` files, scales uint8 -> float32 [0,1],
flattens to [N,784], one-hot encodes labels and splits 60 000 -> 55 000 train +
5 000 validation; ``mnist.train.next_batch(100)`` (``demo1/train.py:154``) walks an
epoch-wise shuffled order and reshuffles at each epoch end.

MI355X design: the host-side reader is only the loader.  ``DeviceDataset`` keeps
the whole split resident in HBM (172 MB fp32 for 55 000 images — nothing next to
288 GB) with an on-device permutation, so the fused conv1 kernel gathers its batch
by index and no host->device copy happens in the step loop.

The reference tree ships without ``train-images-idx3-ubyte.gz``
(``.MISSING_LARGE_BLOBS``), so ``read_data_sets`` can synthesise a train split of
the same shapes (class-conditional stroke prototypes + noise: learnable, so
accuracy curves stay meaningful) or carve one out of the shipped t10k split.
"""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

TRAIN_IMAGES = "train-images-idx3-ubyte.gz"
TRAIN_LABELS = "train-labels-idx1-ubyte.gz"
TEST_IMAGES = "t10k-images-idx3-ubyte.gz"
TEST_LABELS = "t10k-labels-idx1-ubyte.gz"

IMAGE_MAGIC = 2051
LABEL_MAGIC = 2049
NUM_CLASSES = 10
IMAGE_PIXELS = 784


def _open(path: str):
    return gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")


def read_idx_images(path: str) -> np.ndarray:
    """Parse an IDX3 image file -> uint8 [N, rows, cols]."""
    with _open(path) as f:
        magic, n, rows, cols = struct.unpack(">IIII", f.read(16))
        if magic != IMAGE_MAGIC:
            raise ValueError("Invalid magic number %d in MNIST image file: %s" % (magic, path))
        buf = f.read(n * rows * cols)
    if len(buf) != n * rows * cols:
        raise ValueError("truncated IDX image file %s" % path)
    return np.frombuffer(buf, dtype=np.uint8).reshape(n, rows, cols)


def read_idx_labels(path: str) -> np.ndarray:
    """Parse an IDX1 label file -> uint8 [N]."""
    with _open(path) as f:
        magic, n = struct.unpack(">II", f.read(8))
        if magic != LABEL_MAGIC:
            raise ValueError("Invalid magic number %d in MNIST label file: %s" % (magic, path))
        buf = f.read(n)
    if len(buf) != n:
        raise ValueError("truncated IDX label file %s" % path)
    return np.frombuffer(buf, dtype=np.uint8)


def dense_to_one_hot(labels: np.ndarray, num_classes: int = NUM_CLASSES) -> np.ndarray:
    out = np.zeros((labels.shape[0], num_classes), dtype=np.float32)
    out[np.arange(labels.shape[0]), labels.astype(np.int64)] = 1.0
    return out


def _scale_flatten(images_u8: np.ndarray) -> np.ndarray:
    n = images_u8.shape[0]
    return images_u8.reshape(n, -1).astype(np.float32) * np.float32(1.0 / 255.0)


class DataSet:
    """One split with TF-tutorial ``next_batch`` semantics.

    ``images`` float32 [N,784] in [0,1]; ``labels`` one-hot float32 [N,10] (or int
    class ids when ``one_hot=False``).
    """

    def __init__(self, images: np.ndarray, labels: np.ndarray, *, one_hot: bool = True,
                 seed: Optional[int] = None):
        assert images.shape[0] == labels.shape[0], (images.shape, labels.shape)
        self._images = images
        self._labels = labels
        self._one_hot = one_hot
        self._num_examples = images.shape[0]
        self._epochs_completed = 0
        self._index_in_epoch = 0
        self._rng = np.random.RandomState(seed)

    @property
    def images(self) -> np.ndarray:
        return self._images

    @property
    def labels(self) -> np.ndarray:
        return self._labels

    @property
    def num_examples(self) -> int:
        return self._num_examples

    @property
    def epochs_completed(self) -> int:
        return self._epochs_completed

    @property
    def class_ids(self) -> np.ndarray:
        return self._labels.argmax(1).astype(np.int32) if self._one_hot else self._labels.astype(np.int32)

    def _shuffle(self) -> None:
        perm = self._rng.permutation(self._num_examples)
        self._images = self._images[perm]
        self._labels = self._labels[perm]

    def next_batch(self, batch_size: int, shuffle: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """Sequential batches over a shuffled epoch; the tail of one epoch is
        concatenated with the head of the (reshuffled) next one."""
        start = self._index_in_epoch
        if self._epochs_completed == 0 and start == 0 and shuffle:
            self._shuffle()
        if start + batch_size > self._num_examples:
            self._epochs_completed += 1
            rest = self._num_examples - start
            images_rest = self._images[start:self._num_examples]
            labels_rest = self._labels[start:self._num_examples]
            if shuffle:
                self._shuffle()
            start = 0
            self._index_in_epoch = batch_size - rest
            end = self._index_in_epoch
            return (np.concatenate((images_rest, self._images[start:end]), axis=0),
                    np.concatenate((labels_rest, self._labels[start:end]), axis=0))
        self._index_in_epoch += batch_size
        end = self._index_in_epoch
        return self._images[start:end], self._labels[start:end]


@dataclass
class Datasets:
    train: DataSet
    validation: DataSet
    test: DataSet
    source: str = "idx"  # "idx" | "synthetic" | "t10k-split"


# --------------------------------------------------------------------------- synthetic
def synthetic_mnist(n: int, *, seed: int = 0, noise: float = 0.25,
                    num_classes: int = NUM_CLASSES) -> Tuple[np.ndarray, np.ndarray]:
    """Learnable synthetic digits of the MNIST shapes: uint8 [n,28,28], uint8 [n].

    Each class has a fixed smooth stroke prototype (seeded independently of ``seed``
    so train/test draws share classes); samples add a random +-2 px shift, amplitude
    jitter and pixel noise.
    """
    proto_rng = np.random.RandomState(1234567)
    yy, xx = np.mgrid[0:28, 0:28].astype(np.float32)
    protos = np.zeros((num_classes, 28, 28), np.float32)
    for c in range(num_classes):
        img = np.zeros((28, 28), np.float32)
        pts = proto_rng.uniform(6, 22, size=(4, 2)).astype(np.float32)
        for i in range(3):  # three stroke segments rendered as gaussian tubes
            for t in np.linspace(0.0, 1.0, 12, dtype=np.float32):
                cy, cx = pts[i] * (1 - t) + pts[i + 1] * t
                img += np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.3 ** 2))
        protos[c] = np.clip(img / img.max() * 1.4, 0.0, 1.0)
    rng = np.random.RandomState(seed)
    labels = rng.randint(0, num_classes, size=n).astype(np.uint8)
    shifts = rng.randint(-2, 3, size=(n, 2))
    amp = rng.uniform(0.7, 1.0, size=n).astype(np.float32)
    images = np.empty((n, 28, 28), np.uint8)
    chunk = 4096
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        base = protos[labels[s:e]]
        out = np.empty_like(base)
        for j in range(e - s):
            out[j] = np.roll(base[j], tuple(shifts[s + j]), axis=(0, 1))
        out = out * amp[s:e, None, None] + rng.normal(0.0, noise, size=out.shape).astype(np.float32) * 0.5
        images[s:e] = (np.clip(out, 0.0, 1.0) * 255.0).astype(np.uint8)
    return images, labels


def read_data_sets(train_dir: str, one_hot: bool = True, validation_size: int = 5000, *,
                   synthetic: Optional[bool] = None, train_on_t10k: bool = False,
                   seed: Optional[int] = None, synthetic_train_size: int = 60000,
                   synthetic_test_size: int = 10000) -> Datasets:
    """Equivalent of ``input_data.read_data_sets`` (``demo1/train.py:6``).

    ``synthetic=None`` (auto): the real IDX files are used when all four exist; when the train
    images are missing (the shipped tree, ``.MISSING_LARGE_BLOBS``) but t10k is present, the real
    t10k images are split 8 000 train / 1 000 validation / 1 000 test so accuracies are still
    measured on genuine handwriting; with no IDX files at all (or ``synthetic=True``) learnable
    synthetic digits of the same shapes are generated.
    """
    p = lambda name: os.path.join(train_dir, name)  # noqa: E731
    have_test = os.path.exists(p(TEST_IMAGES)) and os.path.exists(p(TEST_LABELS))
    have_train = os.path.exists(p(TRAIN_IMAGES)) and os.path.exists(p(TRAIN_LABELS))
    source = "idx"
    if synthetic is True or (not have_test and not have_train):
        s0 = 0 if seed is None else seed
        tr_i, tr_l = synthetic_mnist(synthetic_train_size, seed=s0 + 1)
        te_i, te_l = synthetic_mnist(synthetic_test_size, seed=s0 + 2)
        source = "synthetic"
    elif have_train and not train_on_t10k:
        tr_i, tr_l = read_idx_images(p(TRAIN_IMAGES)), read_idx_labels(p(TRAIN_LABELS))
        te_i, te_l = read_idx_images(p(TEST_IMAGES)), read_idx_labels(p(TEST_LABELS))
    else:
        all_i, all_l = read_idx_images(p(TEST_IMAGES)), read_idx_labels(p(TEST_LABELS))
        tr_i, tr_l = all_i[:9000], all_l[:9000]
        te_i, te_l = all_i[9000:], all_l[9000:]
        validation_size = min(validation_size, 1000)
        source = "t10k-split"
    if not 0 <= validation_size <= len(tr_i):
        raise ValueError("Validation size should be between 0 and %d. Received: %d."
                         % (len(tr_i), validation_size))
    va_i, va_l = tr_i[:validation_size], tr_l[:validation_size]
    tr_i, tr_l = tr_i[validation_size:], tr_l[validation_size:]

    def mk(i, l, sd):
        lab = dense_to_one_hot(l) if one_hot else l.astype(np.int32)
        return DataSet(_scale_flatten(i), lab, one_hot=one_hot, seed=sd)

    sd = (lambda k: None) if seed is None else (lambda k: seed + k)  # noqa: E731
    return Datasets(train=mk(tr_i, tr_l, sd(11)), validation=mk(va_i, va_l, sd(12)),
                    test=mk(te_i, te_l, sd(13)), source=source)


# --------------------------------------------------------------------------- device resident
class DeviceDataset:
    """A split resident in device memory with an on-device epoch permutation.

    ``images`` fp32 [N,784], ``labels`` int32 [N] live on ``device``.  ``perm`` is a
    device int32 permutation; rank ``r`` of ``world`` consumes ``perm[r::world]`` so
    replicas see disjoint shards under the same seed (the reference samples
    independently and unseeded on every worker, ``demo2/train.py:182``).
    ``batch_indices(step)`` is what the fused conv1 kernel gathers through.
    """

    def __init__(self, split: DataSet, device, *, rank: int = 0, world: int = 1,
                 seed: Optional[int] = None):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.images = torch.from_numpy(np.ascontiguousarray(split.images)).to(self.device)
        self.labels = torch.from_numpy(np.ascontiguousarray(split.class_ids)).to(self.device)
        self.num_examples = split.num_examples
        self.rank, self.world = rank, world
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(int(np.random.SeedSequence(seed).entropy % (2 ** 31)) if seed is None else seed)
        self.shard_size = self.num_examples // world
        self.perm = torch.empty(self.shard_size, dtype=torch.int32, device=self.device)
        self.epoch = -1
        self.reshuffle()

    def reshuffle(self) -> None:
        full = self.torch.randperm(self.num_examples, generator=self._gen)
        shard = full[self.rank::self.world][: self.shard_size].to(self.torch.int32)
        self.perm.copy_(shard.to(self.device), non_blocking=False)
        self.epoch += 1

    def steps_per_epoch(self, batch_size: int) -> int:
        # the fused gather reads perm[(step % spe)*B + b] for every b < B: a batch larger than this rank's shard would
        # read past the permutation (and then past the images), so it is an error, not a short batch
        if batch_size > self.shard_size:
            raise ValueError("batch_size %d exceeds this rank's shard of %d examples (%d examples over %d ranks)"
                             % (batch_size, self.shard_size, self.num_examples, self.world))
        return self.shard_size // batch_size

    def batch_indices(self, step_in_epoch: int, batch_size: int):
        s = step_in_epoch * batch_size
        return self.perm[s:s + batch_size]

    def gather(self, idx):
        """Host-orchestrated gather (eager path / tests): fp32 [B,784], int32 [B]."""
        li = idx.long()
        return self.images.index_select(0, li), self.labels.index_select(0, li)
