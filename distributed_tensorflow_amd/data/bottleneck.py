"""Image preprocessing, batched bottleneck extraction, the HBM-resident cache and the samplers
(components C31-C35, C41).

Reference behaviour (``retrain1/retrain.py``): one Inception forward per image with the raw JPEG fed
into the graph (``run_bottleneck_on_image`` :228-231), results cached as one decimal ``.txt`` per
image (:201-245) that is re-opened and re-parsed for every sample of every batch (:322-340, 409 600
``float()`` calls per training step), class-balanced sampling with replacement (:329-333), and an
optional distortion graph evaluated image by image (:137-165, :300-319).

MI355X design: host threads only decode JPEGs (PIL) into one pinned uint8 buffer; a single HIP
kernel resizes/normalises (and distorts) the whole batch; Inception runs batched in bf16; the
bottlenecks live in one ``[N,2048]`` fp32 tensor in HBM together with label / split index tensors;
sampling is a device kernel that emits row indices which the head kernel gathers through.  Under
data parallelism extraction is sharded over ranks and all-gathered (the reference makes every worker
extract everything, quirk Q9).  ``.txt`` import/export keeps the reference cache format usable.
"""
from __future__ import annotations

import ctypes as C
import os
import random
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import native
from . import image_lists as il
from .image_lists import CATEGORIES, ImageIndex, MAX_NUM_IMAGES_PER_CLASS

BOTTLENECK_SIZE = 2048
BOTTLENECK_TENSOR_SIZE = BOTTLENECK_SIZE  # the reference's name (retrain1/retrain.py:30)
INPUT_SIZE = 299


class ImageDesc(C.Structure):  # csrc/hip/inception_ops.hip: struct ImageDesc
    _fields_ = [("offset", C.c_long), ("H", C.c_int), ("W", C.c_int), ("scale_y", C.c_float), ("scale_x", C.c_float),
                ("off_y", C.c_float), ("off_x", C.c_float), ("flip", C.c_int), ("brightness", C.c_float)]




@dataclass
class Distortion:
    """Flags of ``add_input_distortions`` (``retrain1/retrain.py:137-165``)."""
    flip_left_right: bool = False
    random_crop: int = 0
    random_scale: int = 0
    random_brightness: int = 0

    @property
    def enabled(self) -> bool:  # should_distort_images (:132-134)
        return bool(self.flip_left_right or self.random_crop != 0 or self.random_scale != 0 or self.random_brightness != 0)

    def sample(self, rng: random.Random) -> Tuple[float, float, float, bool, float]:
        """(precrop size, crop offset y, crop offset x, flip, brightness) for one image."""
        margin_scale = 1.0 + self.random_crop / 100.0
        resize_scale = 1.0 + self.random_scale / 100.0
        scale_value = margin_scale * rng.uniform(1.0, resize_scale)
        precrop = int(scale_value * INPUT_SIZE)
        oy = rng.randint(0, max(0, precrop - INPUT_SIZE))
        ox = rng.randint(0, max(0, precrop - INPUT_SIZE))
        flip = self.flip_left_right and rng.random() < 0.5
        bmin, bmax = 1.0 - self.random_brightness / 100.0, 1.0 + self.random_brightness / 100.0
        return float(precrop), float(oy), float(ox), flip, rng.uniform(bmin, bmax)


def decode_jpeg(path: str) -> np.ndarray:
    """File -> uint8 HWC RGB (``DecodeJpeg`` with channels=3)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class ImagePreprocessor:
    """JPEG decode on a host thread pool + one batched resize/normalise(/distort) kernel."""

    def __init__(self, device, workers: int = 8, fp32: bool = False):
        self.device = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.dtype = torch.float32 if fp32 else torch.bfloat16   # fp32: the input of the reference-precision forward

    def decode(self, paths: Sequence[str]) -> List[np.ndarray]:
        return list(self.pool.map(decode_jpeg, paths))

    def batch(self, images: Sequence[np.ndarray], distortion: Optional[Distortion] = None,
              rng: Optional[random.Random] = None) -> torch.Tensor:
        """uint8 HWC arrays (any sizes) -> NHWC ``[B,299,299,8]`` in ``self.dtype`` ((x-128)/128, channels 3..7 zero)."""
        B = len(images)
        descs = (ImageDesc * B)()
        total = sum(int(a.size) for a in images)
        off = 0
        offsets = []
        for i, a in enumerate(images):
            H, W = a.shape[:2]
            d = descs[i]
            d.offset, d.H, d.W = off, H, W
            if distortion is not None and distortion.enabled:
                precrop, oy, ox, flip, bright = distortion.sample(rng or random)
            else:
                precrop, oy, ox, flip, bright = float(INPUT_SIZE), 0.0, 0.0, False, 1.0
            d.scale_y, d.scale_x = H / precrop, W / precrop
            d.off_y, d.off_x, d.flip, d.brightness = oy, ox, int(flip), bright
            offsets.append(off)
            off += int(a.size)
        if self.device.type != "cuda":
            return torch.stack([preprocess_reference(a, descs[i]) for i, a in enumerate(images)]).to(self.dtype)
        host = torch.empty(total, dtype=torch.uint8).pin_memory()
        hv = host.numpy()
        for a, o in zip(images, offsets):
            hv[o:o + a.size] = a.reshape(-1)
        pix = host.to(self.device, non_blocking=True)
        dbytes = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
        out = torch.empty(B, INPUT_SIZE, INPUT_SIZE, 8, dtype=self.dtype, device=self.device)
        lib = native.hip()
        fn = lib.dtf_preprocess_images_f32 if self.dtype == torch.float32 else lib.dtf_preprocess_images
        native.check(fn(pix.data_ptr(), dbytes.data_ptr(), out.data_ptr(), B, INPUT_SIZE,
                        torch.cuda.current_stream().cuda_stream), "preprocess")
        return out

    def load(self, paths: Sequence[str], distortion: Optional[Distortion] = None,
             rng: Optional[random.Random] = None) -> torch.Tensor:
        return self.batch(self.decode(paths), distortion, rng)


def preprocess_reference(img: np.ndarray, d: ImageDesc) -> torch.Tensor:
    """PyTorch statement of ``preprocess_kernel`` for one image -> fp32 ``[299,299,8]``."""
    S = INPUT_SIZE
    src = torch.from_numpy(np.array(img, dtype=np.float32))
    H, W = src.shape[:2]
    ys = torch.arange(S, dtype=torch.float32)
    xs = torch.arange(S, dtype=torch.float32)
    if d.flip:
        xs = (S - 1) - xs
    fy = (ys + np.float32(d.off_y)) * np.float32(d.scale_y)
    fx = (xs + np.float32(d.off_x)) * np.float32(d.scale_x)
    y0 = fy.floor().clamp(max=H - 1).long()
    x0 = fx.floor().clamp(max=W - 1).long()
    y1 = (y0 + 1).clamp(max=H - 1)
    x1 = (x0 + 1).clamp(max=W - 1)
    wy = (fy - y0.float()).view(S, 1, 1)
    wx = (fx - x0.float()).view(1, S, 1)
    p00, p01 = src[y0][:, x0], src[y0][:, x1]
    p10, p11 = src[y1][:, x0], src[y1][:, x1]
    top = p00 + (p01 - p00) * wx
    bot = p10 + (p11 - p10) * wx
    v = ((top + (bot - top) * wy) * np.float32(d.brightness) - 128.0) / 128.0
    out = torch.zeros(S, S, 8)
    out[..., :3] = v
    return out


# ----------------------------------------------------------------------------------- cache
class BottleneckCache:
    """``[N,2048]`` fp32 bottlenecks + label / split tables resident on ``device``."""

    def __init__(self, index: ImageIndex, device):
        self.index = index
        self.device = torch.device(device)
        n = index.num_images
        self.values = torch.zeros(n, BOTTLENECK_SIZE, dtype=torch.float32, device=self.device)
        self.labels = torch.from_numpy(index.class_id.astype(np.int32)).to(self.device)
        self.offsets = torch.from_numpy(index.offsets.astype(np.int64)).to(self.device)  # [C,3]
        self.counts = torch.from_numpy(index.counts.astype(np.int64)).to(self.device)
        self.filled = np.zeros(n, bool)
        self._draws = 0

    # -------------------------------------------------------------- extraction
    def extract(self, backbone, preproc: ImagePreprocessor, batch_size: int = 64, rank: int = 0, world: int = 1,
                txt_dir: Optional[str] = None, progress=print, pipeline=None) -> int:
        """Fill the cache: this rank computes a contiguous 1/world block of the missing rows (batched
        forward), then the blocks are all-gathered.  Returns the number of bottlenecks this rank created.
        ``pipeline``: a ``data.pipeline.DecodePipeline`` (decode processes -> pinned ring -> side-stream H2D, overlapped
        with the forward); without one the batches are decoded, copied and run one after the other."""
        n = self.index.num_images
        if txt_dir:
            self.import_txt(txt_dir)
        per = (n + world - 1) // world
        lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
        todo = [i for i in range(lo, hi) if not self.filled[i]]
        created = 0
        self.extract_stats = {}
        if pipeline is not None and todo:
            rows_t = torch.as_tensor(todo, dtype=torch.long, device=self.device)
            done = [0]

            def consume(first: int, count: int, x: torch.Tensor) -> None:
                feats = backbone(x)
                self.values.index_copy_(0, rows_t[first:first + count], feats[:count].to(torch.float32))
                if (first + count) // 100 != first // 100 and progress:
                    progress(str((first + count) // 100 * 100) + " bottleneck files created.")  # retrain1/retrain.py:179-180
                done[0] = first + count

            in_dtype = getattr(pipeline, "dtype", torch.bfloat16)

            def cpu_batch(imgs, descs):
                return torch.stack([preprocess_reference(a, descs[i]) for i, a in enumerate(imgs)]).to(in_dtype)

            def warm() -> None:  # kernel selection + graph capture at this batch size, under the first decodes
                if self.device.type == "cuda":
                    backbone(torch.zeros(pipeline.batch_size, INPUT_SIZE, INPUT_SIZE, 8, dtype=in_dtype, device=self.device))

            self.extract_stats = pipeline.run([self.index.paths[i] for i in todo], consume, preprocess_cpu=cpu_batch, warm=warm)
            created = done[0]
            todo = []
        for s in range(0, len(todo), batch_size):
            rows = todo[s:s + batch_size]
            x = preproc.load([self.index.paths[i] for i in rows])
            feats = backbone(x)
            self.values[torch.as_tensor(rows, device=self.device)] = feats.to(torch.float32)
            created += len(rows)
            if created // 100 != (created - len(rows)) // 100 and progress:
                progress(str(created // 100 * 100) + " bottleneck files created.")  # retrain1/retrain.py:179-180
        self.filled[lo:hi] = True
        if world > 1:
            import torch.distributed as dist
            # (device tensors on a gloo group — several ranks on one GPU — are staged through the host, see ddp.all_reduce_)
            xdev = "cpu" if dist.get_backend() == "gloo" else self.device
            padded = torch.zeros(per * world, BOTTLENECK_SIZE, dtype=torch.float32, device=xdev)
            local = torch.zeros(per, BOTTLENECK_SIZE, dtype=torch.float32, device=xdev)
            local[:hi - lo] = self.values[lo:hi].to(xdev)
            dist.all_gather_into_tensor(padded, local)
            self.values.copy_(padded[:n])
            self.filled[:] = True
        if txt_dir and rank == 0:
            self.export_txt(txt_dir)
        return created

    # -------------------------------------------------------------- reference .txt compatibility
    def _txt_path(self, txt_dir: str, row: int) -> str:
        p = self.index.paths[row]
        return os.path.join(txt_dir, os.path.basename(os.path.dirname(p)), os.path.basename(p) + ".txt")

    def export_txt(self, txt_dir: str) -> int:
        lib = native.host()
        vals = self.values.cpu().numpy()
        buf = C.create_string_buffer(BOTTLENECK_SIZE * 24)
        n = 0
        for row in range(self.index.num_images):
            path = self._txt_path(txt_dir, row)
            if os.path.exists(path):
                continue
            os.makedirs(os.path.dirname(path), exist_ok=True)
            k = lib.dtf_format_floats(vals[row].ctypes.data_as(C.c_void_p), BOTTLENECK_SIZE, buf, len(buf))
            with open(path, "wb") as f:
                f.write(buf.raw[:k])
            n += 1
        return n

    def import_txt(self, txt_dir: str) -> int:
        """Load existing reference-format files; an invalid float marks the row for re-creation
        (``retrain1/retrain.py:214-221``)."""
        lib = native.host()
        out = np.empty(BOTTLENECK_SIZE, np.float32)
        rows, mats = [], []
        for row in range(self.index.num_images):
            path = self._txt_path(txt_dir, row)
            if self.filled[row] or not os.path.exists(path):
                continue
            text = open(path, "rb").read()
            k = lib.dtf_parse_floats(text, len(text), out.ctypes.data_as(C.c_void_p), BOTTLENECK_SIZE)
            if k != BOTTLENECK_SIZE:
                print("Invalid float found, recreating bottleneck")
                continue
            rows.append(row)
            mats.append(out.copy())
        if rows:
            self.values[torch.as_tensor(rows, device=self.device)] = torch.from_numpy(np.stack(mats)).to(self.device)
            self.filled[rows] = True
        return len(rows)

    # -------------------------------------------------------------- samplers
    def sample_rows(self, category: str, how_many: int, seed: int = 0) -> torch.Tensor:
        """Device int32 row indices.  ``how_many >= 0``: class-balanced with replacement (on a GPU the returned tensor is a
        per-(split, size) buffer, overwritten by the next draw of the same split and size: use it, or copy it, before then);
        ``how_many < 0``: every image of the split in label-major order (``retrain1/retrain.py:341-353``)."""
        s = CATEGORIES.index(category)
        if how_many < 0:
            return torch.from_numpy(self.index.rows_of(category).astype(np.int32)).to(self.device)
        # (per-split offset / count columns and the output buffer are kept: the retrain loop draws twice per step, and the
        # slicing, .contiguous() and torch.empty of a fresh draw cost more host time than the step's kernels take)
        memo = self.__dict__.setdefault("_sample_memo", {})
        if s not in memo:
            memo[s] = (self.offsets[:, s].contiguous(), self.counts[:, s].contiguous(), {})
        offs, cnts, bufs = memo[s]
        self._draws += 1
        if self.device.type == "cuda":
            if how_many not in bufs:
                bufs[how_many] = torch.empty(how_many, dtype=torch.int32, device=self.device)
            rows = bufs[how_many]  # (reused: every consumer is enqueued on the same stream before the next draw)
            native.check(native.hip().dtf_sample_rows(offs.data_ptr(), cnts.data_ptr(), self.index.num_classes, how_many,
                                                      seed, self._draws, rows.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "sample_rows")
            return rows
        g = torch.Generator().manual_seed((seed * 1000003 + self._draws) % (2 ** 63))
        valid = torch.nonzero(cnts > 0).reshape(-1)
        cls = valid[torch.randint(0, len(valid), (how_many,), generator=g)]
        r = torch.randint(0, MAX_NUM_IMAGES_PER_CLASS + 1, (how_many,), generator=g)
        return (offs[cls] + r % cnts[cls]).to(torch.int32)

    def gather(self, rows: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.values.index_select(0, rows.long()), self.labels.index_select(0, rows.long())

    def filenames(self, rows: torch.Tensor) -> List[str]:
        return [self.index.paths[i] for i in rows.cpu().tolist()]


def get_random_distorted_bottlenecks(index: ImageIndex, how_many: int, category: str, backbone,
                                     preproc: ImagePreprocessor, distortion: Distortion, rng: random.Random
                                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """``get_random_distorted_bottlenecks`` (``retrain1/retrain.py:300-319``) as ONE augmentation launch +
    ONE batched forward (the reference runs 2 x how_many batch-1 ``sess.run`` calls)."""
    s = CATEGORIES.index(category)
    C_ = index.num_classes
    paths, labels = [], []
    for _ in range(how_many):
        c = rng.randrange(C_)
        tries = 0
        while index.counts[c, s] == 0 and tries < C_:
            c = (c + 1) % C_
            tries += 1
        image_index = rng.randrange(MAX_NUM_IMAGES_PER_CLASS + 1)
        row = int(index.offsets[c, s] + image_index % max(1, index.counts[c, s]))
        paths.append(index.paths[row])
        labels.append(c)
    x = preproc.load(paths, distortion, rng)
    feats = backbone(x).to(torch.float32)
    return feats, torch.as_tensor(labels, dtype=torch.int32, device=feats.device)


def distorted_bottleneck_stream(index: ImageIndex, how_many: int, category: str, backbone, pipeline, distortion: Distortion,
                                rng: random.Random, steps: int):
    """The same batches as ``steps`` consecutive calls of ``get_random_distorted_bottlenecks`` with this ``rng`` (same
    draws in the same order: per step ``how_many`` x (class, image index), then ``how_many`` distortions), but streamed
    through the decode pipeline (``data/pipeline.py``): the JPEGs of step k+1.. are decoded by the worker processes and
    copied to the device while step k runs.  Yields ``(bottlenecks fp32 [how_many, 2048], labels int32 [how_many])``.
    The plan is drawn up front as flat arrays (row index, label, 5 distortion parameters per image: 28 bytes each)."""
    s = CATEGORIES.index(category)
    C_ = index.num_classes
    n = steps * how_many
    rows = np.empty(n, np.int64)
    labels = np.empty(n, np.int32)
    dist = np.empty((n, 5), np.float64)
    for st in range(steps):
        base = st * how_many
        for j in range(how_many):
            c = rng.randrange(C_)
            tries = 0
            while index.counts[c, s] == 0 and tries < C_:
                c = (c + 1) % C_
                tries += 1
            image_index = rng.randrange(MAX_NUM_IMAGES_PER_CLASS + 1)
            rows[base + j] = int(index.offsets[c, s] + image_index % max(1, index.counts[c, s]))
            labels[base + j] = c
        for j in range(how_many):
            precrop, oy, ox, flip, bright = distortion.sample(rng)
            dist[base + j] = (precrop, oy, ox, float(flip), bright)
    paths = [index.paths[r] for r in rows]
    largest = int(np.ceil(dist[:, 0].max())) if n else INPUT_SIZE
    dev = pipeline.device

    def cpu_batch(imgs, descs):
        return torch.stack([preprocess_reference(a, descs[i]) for i, a in enumerate(imgs)]).to(pipeline.dtype)

    def desc(i: int):
        d = dist[i]
        return d[0], d[1], d[2], d[3] != 0.0, d[4]

    for first, count, x in pipeline.batches(paths, batch_size=how_many, preprocess_cpu=cpu_batch, desc_fn=desc,
                                            draft=max(2 * INPUT_SIZE, largest)):
        feats = backbone(x).to(torch.float32)
        yield feats[:count], torch.from_numpy(labels[first:first + count]).to(dev)


# ----------------------------------------------------------------------------- packed-double files (C44)
def write_list_of_floats_to_file(list_of_floats, file_path: str) -> None:
    """``retrain1/retrain.py:357-360``: ``BOTTLENECK_TENSOR_SIZE`` native doubles, ``struct.pack('d'*2048)``.
    Never called by the reference's own drivers; kept as the compact binary export of one cache row."""
    a = np.asarray(list_of_floats, dtype=np.float64).reshape(-1)
    if a.size != BOTTLENECK_TENSOR_SIZE:
        raise ValueError("expected %d values, got %d" % (BOTTLENECK_TENSOR_SIZE, a.size))
    with open(file_path, "wb") as f:
        f.write(a.tobytes())


def read_list_of_floats_from_file(file_path: str) -> List[float]:
    """``retrain1/retrain.py:363-366``."""
    with open(file_path, "rb") as f:
        raw = f.read()
    if len(raw) != 8 * BOTTLENECK_TENSOR_SIZE:
        raise ValueError("%s: %d bytes, expected %d" % (file_path, len(raw), 8 * BOTTLENECK_TENSOR_SIZE))
    return np.frombuffer(raw, dtype=np.float64).tolist()

