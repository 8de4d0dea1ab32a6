"""Image list builder with the deterministic SHA-1 split (components C29, C30).

Behavioural parity with ``create_image_lists`` / ``get_image_path``
(``retrain1/retrain.py:78-128,184-197``):

* sub-directories of ``image_dir`` are classes; extensions ``jpg jpeg JPG JPEG``;
* label name = directory name lower-cased with runs of non ``[a-z0-9]`` -> one space;
* split: SHA-1 of the *full path string* with ``_nohash_.*`` stripped (Q18: moving
  ``image_dir`` reshuffles the split), ``% (2**27)`` scaled to a percentage;
  ``< validation_pct`` -> validation, ``< testing_pct + validation_pct`` -> testing,
  else training;
* warnings for < 20 or > 2**27-1 images per class.

Design differences: a missing directory raises a clear error instead of returning
``None`` and crashing later (Q19); an ``ImageIndex`` flattens the lists into integer
arrays (class id, split id, offsets) so the bottleneck cache and the device-side
sampler work on tensors, not on python dicts.
"""
from __future__ import annotations

import glob
import hashlib
import os
import re
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

MAX_NUM_IMAGES_PER_CLASS = 2 ** 27 - 1  # ~134M (retrain1/retrain.py:36)
CATEGORIES = ("training", "testing", "validation")
EXTENSIONS = ("jpg", "jpeg", "JPG", "JPEG")


class ImageDirError(RuntimeError):
    pass


def percentage_hash(file_name: str) -> float:
    """The reference's split hash (``retrain1/retrain.py:111-115``)."""
    hash_name = re.sub(r"_nohash_.*$", "", file_name)
    digest = hashlib.sha1(hash_name.encode("utf-8")).hexdigest()
    return (int(digest, 16) % (MAX_NUM_IMAGES_PER_CLASS + 1)) * (100.0 / MAX_NUM_IMAGES_PER_CLASS)


def label_from_dir(dir_name: str) -> str:
    return re.sub(r"[^a-z0-9]+", " ", dir_name.lower())


def create_image_lists(image_dir: str, testing_percentage: int, validation_percentage: int,
                       *, verbose: bool = True) -> "OrderedDict[str, Dict[str, object]]":
    if not os.path.exists(image_dir):
        raise ImageDirError("Image directory '" + image_dir + "' not found.")
    result: "OrderedDict[str, Dict[str, object]]" = OrderedDict()
    sub_dirs = [x[0] for x in os.walk(image_dir)]
    for sub_dir in sorted(sub_dirs[1:]):  # sorted: the label order (= class ids) must not depend on the filesystem  # first entry is the root itself
        dir_name = os.path.basename(sub_dir)
        if dir_name == image_dir:
            continue
        if verbose:
            print("Looking for images in '" + dir_name + "'")
        file_list: List[str] = []
        for extension in EXTENSIONS:
            file_list.extend(sorted(glob.glob(os.path.join(image_dir, dir_name, "*." + extension))))
        if not file_list:
            if verbose:
                print("No files found")
            continue
        if len(file_list) < 20:
            if verbose:
                print("WARNING: Folder has less than 20 images, which may cause issues.")
        elif len(file_list) > MAX_NUM_IMAGES_PER_CLASS:
            print("WARNING: Folder {} has more than {} images. Some images will never be selected."
                  .format(dir_name, MAX_NUM_IMAGES_PER_CLASS))
        training, testing, validation = [], [], []
        for file_name in file_list:
            base_name = os.path.basename(file_name)
            pct = percentage_hash(file_name)
            if pct < validation_percentage:
                validation.append(base_name)
            elif pct < (testing_percentage + validation_percentage):
                testing.append(base_name)
            else:
                training.append(base_name)
        result[label_from_dir(dir_name)] = {
            "dir": dir_name, "training": training, "testing": testing, "validation": validation,
        }
    return result


def get_image_path(image_lists, label_name: str, index: int, image_dir: str, category: str) -> str:
    """``index % len(list)`` lookup (``retrain1/retrain.py:184-197``); errors are exceptions."""
    if label_name not in image_lists:
        raise KeyError("Label does not exist %s." % label_name)
    label_lists = image_lists[label_name]
    if category not in label_lists:
        raise KeyError("Category does not exist %s." % category)
    category_list = label_lists[category]
    if not category_list:
        raise ValueError("Label %s has no images in the category %s." % (label_name, category))
    base_name = category_list[index % len(category_list)]
    return os.path.join(image_dir, label_lists["dir"], base_name)


def get_bottleneck_path(image_lists, label_name, index, bottleneck_dir, category) -> str:
    return get_image_path(image_lists, label_name, index, bottleneck_dir, category) + ".txt"


@dataclass
class ImageIndex:
    """Flat, tensor-friendly view of ``image_lists``.

    Rows are ordered label -> category (training, testing, validation) -> list order,
    i.e. exactly the order in which the reference's ``cache_bottlenecks`` walks them
    (``retrain1/retrain.py:172-175``).  ``offsets[c, s]`` / ``counts[c, s]`` locate the
    contiguous block of class ``c`` split ``s``.
    """

    labels: List[str]
    paths: List[str]
    class_id: np.ndarray   # int32 [N]
    split_id: np.ndarray   # int32 [N]  (index into CATEGORIES)
    offsets: np.ndarray    # int64 [C, 3]
    counts: np.ndarray     # int64 [C, 3]

    @property
    def num_images(self) -> int:
        return len(self.paths)

    @property
    def num_classes(self) -> int:
        return len(self.labels)

    def rows_of(self, category: str) -> np.ndarray:
        """All rows of a split in reference order (label-major) — the ``how_many < 0`` path."""
        s = CATEGORIES.index(category)
        parts = [np.arange(self.offsets[c, s], self.offsets[c, s] + self.counts[c, s])
                 for c in range(self.num_classes)]
        return np.concatenate(parts).astype(np.int64) if parts else np.zeros((0,), np.int64)


def build_image_index(image_lists, image_dir: str) -> ImageIndex:
    labels = list(image_lists.keys())
    paths: List[str] = []
    cls: List[int] = []
    spl: List[int] = []
    C = len(labels)
    offsets = np.zeros((C, 3), np.int64)
    counts = np.zeros((C, 3), np.int64)
    for ci, label in enumerate(labels):
        ll = image_lists[label]
        for si, cat in enumerate(CATEGORIES):
            offsets[ci, si] = len(paths)
            counts[ci, si] = len(ll[cat])
            for base in ll[cat]:
                paths.append(os.path.join(image_dir, ll["dir"], base))
                cls.append(ci)
                spl.append(si)
    return ImageIndex(labels, paths, np.asarray(cls, np.int32), np.asarray(spl, np.int32), offsets, counts)


def make_synthetic_image_dir(root: str, *, num_classes: int = 5, images_per_class: int = 40,
                             size: Tuple[int, int] = (320, 240), seed: int = 0,
                             quality: int = 90) -> str:
    """Write a separable synthetic JPEG dataset ``root/<class>/*.jpg`` (BASELINE config 3:
    "5-class 299x299 synthetic").  Each class is a distinct colour/texture pattern + noise."""
    from PIL import Image

    rng = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    w, h = size
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for c in range(num_classes):
        d = os.path.join(root, "class_%02d" % c)
        os.makedirs(d, exist_ok=True)
        base_col = np.array([(c * 53) % 256, (c * 97 + 60) % 256, (c * 151 + 120) % 256], np.float32)
        freq = 0.02 + 0.015 * c
        ang = c * 0.7
        for i in range(images_per_class):
            phase = rng.uniform(0, 2 * np.pi)
            wave = np.sin((xx * np.cos(ang) + yy * np.sin(ang)) * freq * 2 * np.pi + phase)
            img = base_col[None, None, :] * (0.6 + 0.4 * wave[..., None])
            img += rng.normal(0, 12.0, size=img.shape)
            Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(
                os.path.join(d, "img_%04d.jpg" % i), quality=quality)
    return root
