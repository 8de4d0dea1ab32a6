import os

import numpy as np
import pytest

from distributed_tensorflow_amd.data import image_lists as il
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.data.imageprepare import imageprepare, prepare_batch, walk_images


def test_idx_parser_on_shipped_t10k(assets_dir):
    d = os.path.join(assets_dir, "MNIST_data")
    imgs = mnist.read_idx_images(os.path.join(d, mnist.TEST_IMAGES))
    labels = mnist.read_idx_labels(os.path.join(d, mnist.TEST_LABELS))
    assert imgs.shape == (10000, 28, 28) and imgs.dtype == np.uint8
    assert labels.shape == (10000,) and labels[0] == 7 and labels[1] == 2  # canonical first test labels
    train_labels = mnist.read_idx_labels(os.path.join(d, mnist.TRAIN_LABELS))
    assert train_labels.shape == (60000,) and train_labels[0] == 5
    with pytest.raises(ValueError):
        mnist.read_idx_images(os.path.join(d, mnist.TEST_LABELS))


def test_read_data_sets_falls_back_to_t10k_split(assets_dir):
    ds = mnist.read_data_sets(os.path.join(assets_dir, "MNIST_data"), one_hot=True, seed=3)
    assert ds.source == "t10k-split"
    assert ds.train.num_examples == 8000 and ds.validation.num_examples == 1000 and ds.test.num_examples == 1000
    assert ds.train.images.shape == (8000, 784) and ds.train.images.dtype == np.float32
    assert 0.0 <= ds.train.images.min() and ds.train.images.max() <= 1.0
    assert ds.train.labels.shape == (8000, 10) and np.all(ds.train.labels.sum(1) == 1)


def test_synthetic_shapes_and_determinism(tmp_path):
    ds = mnist.read_data_sets(str(tmp_path), synthetic=True, seed=1, synthetic_train_size=600,
                              synthetic_test_size=100, validation_size=50)
    assert ds.source == "synthetic" and ds.train.num_examples == 550 and ds.test.num_examples == 100
    a, la = mnist.synthetic_mnist(64, seed=5)
    b, lb = mnist.synthetic_mnist(64, seed=5)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    assert set(np.unique(la)) <= set(range(10))


def test_next_batch_epoch_semantics():
    imgs = np.arange(10, dtype=np.float32).reshape(10, 1)
    labels = mnist.dense_to_one_hot(np.arange(10) % 10)
    ds = mnist.DataSet(imgs, labels, seed=0)
    seen = []
    for _ in range(3):  # 3 x 4 = 12 > 10: crosses the epoch boundary once
        x, y = ds.next_batch(4)
        assert x.shape == (4, 1) and y.shape == (4, 10)
        seen.append(x[:, 0])
    first_epoch = np.concatenate(seen)[:10]
    assert sorted(first_epoch.tolist()) == list(range(10))  # every example exactly once per epoch
    assert ds.epochs_completed == 1


def test_device_dataset_shards_are_disjoint():
    imgs, labels = mnist.synthetic_mnist(200, seed=0)
    split = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels))
    a = mnist.DeviceDataset(split, "cpu", rank=0, world=2, seed=7)
    b = mnist.DeviceDataset(split, "cpu", rank=1, world=2, seed=7)
    sa, sb = set(a.perm.tolist()), set(b.perm.tolist())
    assert len(sa) == 100 and len(sb) == 100 and not (sa & sb)
    x, y = a.gather(a.batch_indices(0, 10))
    assert x.shape == (10, 784) and y.dtype.is_floating_point is False


def test_imageprepare_on_shipped_digits(assets_dir):
    files = walk_images(os.path.join(assets_dir, "digit_imgs"))
    assert len(files) == 6
    for f in files:
        v = imageprepare(f)
        assert len(v) == 784 and 0.0 <= min(v) and max(v) <= 1.0
    batch = prepare_batch(files)
    assert batch.shape == (6, 784) and batch.dtype == np.float32
    # white background -> mostly zeros, ink present
    assert (batch < 0.1).mean() > 0.5 and batch.max() > 0.5


def _make_tree(root, classes=("Rose Bush", "daisy_x"), n=30):
    from PIL import Image
    for c in classes:
        os.makedirs(os.path.join(root, c), exist_ok=True)
        for i in range(n):
            Image.new("RGB", (8, 8), (i, i, i)).save(os.path.join(root, c, "im_%03d.jpg" % i))
    return root


def test_create_image_lists_split_is_deterministic(tmp_path):
    root = _make_tree(str(tmp_path / "imgs"))
    l1 = il.create_image_lists(root, 10, 10, verbose=False)
    l2 = il.create_image_lists(root, 10, 10, verbose=False)
    assert l1 == l2
    assert list(l1.keys()) == sorted(l1.keys()) or set(l1.keys()) == {"rose bush", "daisy x"}
    assert set(l1.keys()) == {"rose bush", "daisy x"}  # lower-cased, non-alnum -> space
    for lab, d in l1.items():
        assert len(d["training"]) + len(d["testing"]) + len(d["validation"]) == 30
        # hash is over the full path with _nohash_ stripped
        for base in d["validation"]:
            assert il.percentage_hash(os.path.join(root, d["dir"], base)) < 10
    assert il.percentage_hash("/a/b_nohash_1.jpg") == il.percentage_hash("/a/b_nohash_2.jpg")
    with pytest.raises(il.ImageDirError):
        il.create_image_lists(str(tmp_path / "missing"), 10, 10)


def test_get_image_path_and_index(tmp_path):
    root = _make_tree(str(tmp_path / "imgs"))
    lists = il.create_image_lists(root, 10, 10, verbose=False)
    lab = list(lists.keys())[0]
    n = len(lists[lab]["training"])
    p0 = il.get_image_path(lists, lab, 0, root, "training")
    assert il.get_image_path(lists, lab, n, root, "training") == p0  # index % len
    assert il.get_bottleneck_path(lists, lab, 0, "/bn", "training").endswith(".jpg.txt")
    with pytest.raises(KeyError):
        il.get_image_path(lists, "nope", 0, root, "training")
    idx = il.build_image_index(lists, root)
    assert idx.num_images == 60 and idx.num_classes == 2
    assert idx.counts.sum() == 60
    rows = idx.rows_of("training")
    assert np.all(idx.split_id[rows] == 0)


def test_synthetic_image_dir(tmp_path):
    root = il.make_synthetic_image_dir(str(tmp_path / "syn"), num_classes=3, images_per_class=4, size=(64, 48))
    lists = il.create_image_lists(root, 10, 10, verbose=False)
    assert len(lists) == 3


def test_packed_double_bottleneck_files(tmp_path):
    from distributed_tensorflow_amd.data import bottleneck as bn
    v = np.random.RandomState(0).randn(2048)
    p = str(tmp_path / "row.bin")
    bn.write_list_of_floats_to_file(v.tolist(), p)
    import struct
    raw = open(p, "rb").read()
    assert struct.unpack("d" * 2048, raw) == tuple(v.tolist())  # the reference's exact encoding
    assert bn.read_list_of_floats_from_file(p) == v.tolist()
    with pytest.raises(ValueError):
        bn.write_list_of_floats_to_file([1.0, 2.0], p)


def test_decode_pipeline_matches_serial_extraction(tmp_path):
    """The extraction pipeline (decode processes -> shared ring -> preprocess) fills the cache with exactly the bottlenecks
    the serial path computes, including a short last batch and images of mixed sizes, one of them large enough for the
    reduced-scale (draft) decode."""
    import numpy as np
    import torch
    from PIL import Image
    from distributed_tensorflow_amd.models.retrain_head import TinyBackbone
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.bottleneck import BottleneckCache, ImagePreprocessor
    from distributed_tensorflow_amd.data.pipeline import DecodePipeline
    d = il.make_synthetic_image_dir(str(tmp_path / "imgs"), num_classes=2, images_per_class=11, seed=4)
    yy, xx = np.mgrid[0:1300, 0:1500].astype(np.float32)   # a smooth photograph-like image (noise would alias)
    big = np.stack([127 + 120 * np.sin(yy / 90.0), 127 + 120 * np.cos(xx / 70.0), (yy + xx) / 2800.0 * 255], -1).astype(np.uint8)
    sub = sorted(os.listdir(d))[0]
    Image.fromarray(big).save(os.path.join(d, sub, "zz_big.jpg"), quality=90)
    lists = il.create_image_lists(d, 10, 10)
    index = il.build_image_index(lists, d)
    backbone = TinyBackbone("cpu", seed=1)
    a = BottleneckCache(index, "cpu")
    a.extract(backbone, ImagePreprocessor("cpu"), batch_size=8, progress=None)
    pipe = DecodePipeline("cpu", batch_size=8, workers=2, slots=3, chunk=3)
    try:
        b = BottleneckCache(index, "cpu")
        n = b.extract(backbone, ImagePreprocessor("cpu"), batch_size=8, progress=None, pipeline=pipe)
        st = b.extract_stats
    finally:
        pipe.close()
    assert n == index.num_images == 23 and st["batches"] == 3 and st["images"] == 23
    # the large image was decoded at 1/2 scale (650 x 750): fewer bytes than the full decode
    assert st["h2d_bytes"] < 22 * 240 * 320 * 3 + 1300 * 1500 * 3 // 2
    small = [i for i, p in enumerate(index.paths) if not p.endswith("zz_big.jpg")]
    assert torch.equal(a.values[small], b.values[small])
    bigrow = [i for i, p in enumerate(index.paths) if p.endswith("zz_big.jpg")]
    # draft decode = DCT-domain 2x reduction before the bilinear resample: close to, not identical with, the full decode
    rel = float((a.values[bigrow] - b.values[bigrow]).norm() / a.values[bigrow].norm())
    assert rel < 0.05, rel
    assert "images/s" in DecodePipeline.describe(st)


def test_decode_pipeline_reports_undecodable_files(tmp_path):
    """A file that is not an image fails the header probe with a clear error before any batch is scheduled."""
    import pytest
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.pipeline import DecodePipeline
    d = il.make_synthetic_image_dir(str(tmp_path / "imgs"), num_classes=1, images_per_class=3, seed=1)
    sub = sorted(os.listdir(d))[0]
    bad = os.path.join(d, sub, "broken.jpg")
    open(bad, "wb").write(b"not a jpeg at all")
    paths = sorted(os.path.join(d, sub, f) for f in os.listdir(os.path.join(d, sub)))
    pipe = DecodePipeline("cpu", batch_size=2, workers=1, slots=2, chunk=2)
    try:
        with pytest.raises(ValueError, match="undecodable"):
            pipe.run(paths, lambda first, count, x: None, preprocess_cpu=lambda imgs, descs: None)
    finally:
        pipe.close()


def test_distorted_stream_equals_per_step_distorted_bottlenecks(tmp_path):
    """``distorted_bottleneck_stream`` (decode pipeline) yields, step for step, what consecutive calls of
    ``get_random_distorted_bottlenecks`` yield from the same random state (reference: retrain1/retrain.py:300-319)."""
    import random
    import torch
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.bottleneck import (Distortion, ImagePreprocessor, distorted_bottleneck_stream,
                                                            get_random_distorted_bottlenecks)
    from distributed_tensorflow_amd.data.pipeline import DecodePipeline
    from distributed_tensorflow_amd.models.retrain_head import TinyBackbone
    d = il.make_synthetic_image_dir(str(tmp_path / "imgs"), num_classes=3, images_per_class=12, seed=2)
    index = il.build_image_index(il.create_image_lists(d, 10, 10), d)
    backbone = TinyBackbone("cpu", seed=1)
    dist = Distortion(flip_left_right=True, random_crop=10, random_scale=20, random_brightness=15)
    steps, B = 3, 6
    rng_a = random.Random(42)
    ref = [get_random_distorted_bottlenecks(index, B, "training", backbone, ImagePreprocessor("cpu"), dist, rng_a)
           for _ in range(steps)]
    pipe = DecodePipeline("cpu", batch_size=4, workers=2, slots=3, chunk=2)
    try:
        got = list(distorted_bottleneck_stream(index, B, "training", backbone, pipe, dist, random.Random(42), steps))
    finally:
        pipe.close()
    assert len(got) == steps and pipe.stats["batches"] == steps
    for (fa, la), (fb, lb) in zip(ref, got):
        assert torch.equal(la.cpu(), lb.cpu()) and fa.shape == fb.shape == (B, 2048)
        assert torch.equal(fa, fb)
    assert len({tuple(l.tolist()) for _, l in got}) > 1   # (the steps really differ)
