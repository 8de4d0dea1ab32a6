"""TensorBoard summaries (components C18, C24).

``FileWriter`` wraps the native tfevents writer (``csrc/host/tfevents_writer.cc``): records are
framed and CRC'd in C++ and flushed by a background thread, so ``add_summary`` never blocks the
step loop (the reference evaluates and writes summaries synchronously on *every* step,
``demo1/train.py:155-157`` / ``retrain1/retrain.py:440-446`` — quirk Q5).  A JSONL mirror of the
scalars is written next to the event file for tooling that has no TensorBoard (it is not
installed here either).

``variable_summaries`` reproduces ``retrain1/retrain.py:249-258`` (mean / stddev / max / min +
histogram per variable) from one fused single-pass statistics kernel on the GPU
(``csrc/hip/stats.hip``) or torch ops on CPU.  Histograms use TensorFlow's default exponential
bucket limits so TensorBoard renders them the same way.

``read_events`` re-parses an event file (length + CRC checks and a minimal protobuf decoder):
the test-side inverse of the writer.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import socket
import struct
import threading
import time
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from .. import native


def _default_bucket_limits() -> np.ndarray:
    """TF's ``histogram::InitDefaultBuckets``: +-1e-12 * 1.1^k up to 1e20, plus DBL_MAX."""
    pos = []
    v = 1e-12
    while v < 1e20:
        pos.append(v)
        v *= 1.1
    pos = np.asarray(pos, np.float64)
    return np.concatenate((-pos[::-1], [0.0], pos, [np.finfo(np.float64).max]))


BUCKET_LIMITS = _default_bucket_limits()


class Summary:
    """A bag of values for one step (the analogue of the serialized ``Summary`` proto)."""

    def __init__(self):
        self.scalars: List[Tuple[str, float]] = []
        self.histograms: List[Tuple[str, dict]] = []

    def scalar(self, tag: str, value) -> "Summary":
        self.scalars.append((tag, float(value)))
        return self

    def histogram(self, tag: str, values) -> "Summary":
        self.histograms.append((tag, histogram_stats(values)))
        return self

    def merge(self, other: "Summary") -> "Summary":
        self.scalars += other.scalars
        self.histograms += other.histograms
        return self


_NPOS = (len(BUCKET_LIMITS) - 2) // 2
_DEV_LIMITS: dict = {}


def _order_key_decode(k: int) -> float:
    u = (~k & 0xFFFFFFFF) if not (k & 0x80000000) else (k & 0x7FFFFFFF)
    return struct.unpack("<f", struct.pack("<I", u))[0]


def device_stats(t) -> Tuple[float, float, float, float, float, np.ndarray]:
    """One launch of the fused stats kernel (``csrc/hip/stats.hip``) over a CUDA fp32 tensor ->
    (min, max, num, sum, sum_squares, bucket counts)."""
    import torch
    t = t.detach().reshape(-1)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    dev = t.device
    lim = _DEV_LIMITS.get(dev)
    if lim is None:
        lim = torch.from_numpy(BUCKET_LIMITS[_NPOS + 1:_NPOS + 1 + _NPOS].copy()).to(dev)
        _DEV_LIMITS[dev] = lim
    sums = torch.empty(2, dtype=torch.float64, device=dev)
    counts = torch.empty(2 * _NPOS + 4, dtype=torch.int32, device=dev)
    native.check(native.hip().dtf_tensor_stats(t.data_ptr(), t.numel(), lim.data_ptr(), _NPOS, sums.data_ptr(),
                                               counts.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                 "tensor_stats")
    c = counts.cpu().numpy().view(np.uint32)
    sm = sums.cpu().numpy()
    nb = 2 * _NPOS + 2
    hist = c[:nb].astype(np.int64)
    num = float(hist.sum())
    mn = _order_key_decode(int(c[nb])) if num else 0.0
    mx = _order_key_decode(int(c[nb + 1])) if num else 0.0
    return mn, mx, num, float(sm[0]), float(sm[1]), hist


class StatsRing:
    """Per-step tensor statistics without a host sync per step: ``record(step, k, tensor)`` runs the fused statistics
    kernel (``csrc/hip/stats.hip``) into slot ``step % slots`` of a device ring, ``flush(first, last)`` reads the ring
    back once and returns, per step, the same ``Summary`` entries ``variable_summaries`` / ``Summary.histogram`` would
    have produced.  ``tags[k] = (tag, with_scalars)``: ``with_scalars`` adds the mean / stddev / max / min scalars of
    ``variable_summaries`` (``retrain1/retrain.py:249-258``), otherwise the entry is a plain histogram (``:277,279``).
    The reference evaluates all of them on every step for the train and the validation batch (``:440-446``)."""

    def __init__(self, tags, device, slots: int = 256):
        import torch
        self.tags, self.slots, self.device = list(tags), slots, torch.device(device)
        self.T = len(self.tags)
        self.ncnt = 2 * _NPOS + 4
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            self.sums = torch.zeros(slots, self.T, 2, dtype=torch.float64, device=self.device)
            self.counts = torch.zeros(slots, self.T, self.ncnt, dtype=torch.int32, device=self.device)
            lim = _DEV_LIMITS.get(self.device)
            if lim is None:
                lim = torch.from_numpy(BUCKET_LIMITS[_NPOS + 1:_NPOS + 1 + _NPOS].copy()).to(self.device)
                _DEV_LIMITS[self.device] = lim
            self._lim = lim
        else:
            self._host = {}

    def record(self, step: int, k: int, t) -> None:
        import torch
        slot = step % self.slots
        if not self.cuda:
            self._host[(slot, k)] = histogram_stats(t)
            return
        t = t.detach().reshape(-1)
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        native.check(native.hip().dtf_tensor_stats(t.data_ptr(), t.numel(), self._lim.data_ptr(), _NPOS,
                                                   self.sums[slot, k].data_ptr(), self.counts[slot, k].data_ptr(),
                                                   torch.cuda.current_stream(self.device).cuda_stream), "tensor_stats")

    def flush(self, first: int, last: int, only=None) -> Dict[int, Summary]:
        """Summaries of steps [first, last) (at most ``slots`` back); ``only``: indices into ``tags`` to include."""
        assert last - first <= self.slots
        out: Dict[int, Summary] = {}
        if last <= first:
            return out
        if self.cuda:
            sums = self.sums.cpu().numpy()
            counts = self.counts.cpu().numpy().view(np.uint32)
        nb = 2 * _NPOS + 2
        for step in range(first, last):
            slot = step % self.slots
            s = Summary()
            for k, (tag, with_scalars) in enumerate(self.tags):
                if only is not None and k not in only:
                    continue
                if self.cuda:
                    c = counts[slot, k]
                    hist = c[:nb].astype(np.int64)
                    num = float(hist.sum())
                    mn = _order_key_decode(int(c[nb])) if num else 0.0
                    mx = _order_key_decode(int(c[nb + 1])) if num else 0.0
                    h = _compress(mn, mx, num, float(sums[slot, k, 0]), float(sums[slot, k, 1]), hist)
                else:
                    h = self._host.get((slot, k))
                    if h is None:
                        continue
                if with_scalars:
                    num = max(h["num"], 1.0)
                    mean = h["sum"] / num
                    s.scalar(tag + "/summaries/mean", mean)
                    s.scalar(tag + "/summaries/stddev/stddev", max(h["sum_squares"] / num - mean * mean, 0.0) ** 0.5)
                    s.scalar(tag + "/summaries/max", h["max"])
                    s.scalar(tag + "/summaries/min", h["min"])
                    s.histograms.append((tag + "/summaries/histogram", h))
                else:
                    s.histograms.append((tag, h))
            out[step] = s
        return out


def histogram_stats(values) -> dict:
    """min/max/num/sum/sum_squares + TF-style compressed buckets of a tensor or array.  Bucket ``i`` holds
    ``limit[i-1] <= v < limit[i]`` (TF's ``Histogram::Add`` uses ``upper_bound``)."""
    try:
        import torch
        if isinstance(values, torch.Tensor):
            if values.is_cuda:
                return _compress(*device_stats(values))
            values = values.detach().reshape(-1).to(torch.float64).cpu().numpy()
    except ImportError:
        pass
    a = np.asarray(values, np.float64).reshape(-1)
    a = a[~np.isnan(a)]
    idx = np.searchsorted(BUCKET_LIMITS, a, side="right")
    counts = np.bincount(np.minimum(idx, len(BUCKET_LIMITS) - 1), minlength=len(BUCKET_LIMITS))
    if a.size == 0:
        return _compress(0.0, 0.0, 0.0, 0.0, 0.0, counts)
    return _compress(float(a.min()), float(a.max()), float(a.size), float(a.sum()), float((a * a).sum()), counts)


def _compress(mn, mx, num, s, ss, counts) -> dict:
    """Drop runs of empty buckets like TF's ``EncodeToProto`` does."""
    limits, buckets = [], []
    n = len(counts)
    i = 0
    while i < n:
        j = i
        while j < n and counts[j] == 0:
            j += 1
        if j > i and j < n:
            limits.append(float(BUCKET_LIMITS[j - 1]))
            buckets.append(0.0)
        elif j > i:
            break
        i = j
        if i < n:
            limits.append(float(BUCKET_LIMITS[i]))
            buckets.append(float(counts[i]))
            i += 1
    if not limits:
        limits, buckets = [float(BUCKET_LIMITS[-1])], [0.0]
    return {"min": mn, "max": mx, "num": num, "sum": s, "sum_squares": ss,
            "bucket_limit": limits, "bucket": buckets}


def variable_summaries(name: str, var) -> Summary:
    """``variable_summaries(var)`` of the reference under name scope ``name``
    (``retrain1/retrain.py:249-258``): mean, stddev, max, min scalars + histogram — all five from
    one pass over the tensor (one kernel launch + one small read-back on the GPU)."""
    import torch
    s = Summary()
    if isinstance(var, torch.Tensor) and var.is_cuda:
        mn, mx, num, sm, ss, hist = device_stats(var)
        h = _compress(mn, mx, num, sm, ss, hist)
    else:
        h = histogram_stats(var)
        mn, mx, num, sm, ss = h["min"], h["max"], h["num"], h["sum"], h["sum_squares"]
    mean = sm / max(num, 1.0)
    std = max(ss / max(num, 1.0) - mean * mean, 0.0) ** 0.5
    s.scalar(name + "/summaries/mean", mean)
    s.scalar(name + "/summaries/stddev/stddev", std)
    s.scalar(name + "/summaries/max", mx)
    s.scalar(name + "/summaries/min", mn)
    s.histograms.append((name + "/summaries/histogram", h))
    return s


class FileWriter:
    """``tf.summary.FileWriter(logdir, graph)`` equivalent (``demo1/train.py:151``)."""

    def __init__(self, logdir: str, graph: Optional[dict] = None, filename_suffix: str = ""):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        fname = "events.out.tfevents.%010d.%s%s" % (int(time.time()), socket.gethostname(), filename_suffix)
        self.path = os.path.join(logdir, fname)
        self._lib = native.host()
        self._h = self._lib.dtf_events_open(self.path.encode())
        if not self._h:
            raise OSError("cannot open event file " + self.path)
        self._jsonl = open(os.path.join(logdir, "scalars%s.jsonl" % filename_suffix), "a")
        # One Summary is assembled in the native writer's per-handle state between begin and end: the training thread and
        # the Supervisor's autosave thread (global_step/sec) both write through this object, so a whole add_summary is one
        # critical section (ctypes releases the GIL during the native calls).
        self._lock = threading.Lock()
        if graph is not None:
            blob = json.dumps(graph).encode()
            self._lib.dtf_events_graph(self._h, blob, len(blob))

    def add_summary(self, summary: Summary, global_step: int, wall_time: Optional[float] = None) -> None:
        lib = self._lib
        with self._lock:
            h = self._h
            if not h:
                return  # closed (a late write from a service thread during shutdown)
            lib.dtf_events_begin(h, int(global_step), float(wall_time or 0.0))
            for tag, v in summary.scalars:
                lib.dtf_events_scalar(h, tag.encode(), float(v))
            for tag, hs in summary.histograms:
                n = len(hs["bucket_limit"])
                lim = (C.c_double * n)(*hs["bucket_limit"])
                cnt = (C.c_double * n)(*hs["bucket"])
                lib.dtf_events_histogram(h, tag.encode(), hs["min"], hs["max"], hs["num"], hs["sum"], hs["sum_squares"],
                                         n, lim, cnt)
            lib.dtf_events_end(h)
            if summary.scalars:
                self._jsonl.write(json.dumps({"step": int(global_step), **{t: v for t, v in summary.scalars}}) + "\n")

    def add_scalars(self, values: Dict[str, float], global_step: int) -> None:
        s = Summary()
        for k, v in values.items():
            s.scalar(k, v)
        self.add_summary(s, global_step)

    def flush(self) -> None:
        with self._lock:
            if self._h:
                self._lib.dtf_events_flush(self._h)
                self._jsonl.flush()

    def close(self) -> None:
        with self._lock:
            if self._h:
                self._lib.dtf_events_close(self._h)
                self._h = None
                self._jsonl.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------- reader (tests / tooling)
def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _parse_fields(buf: bytes) -> List[Tuple[int, int, object]]:
    out = []
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wire)
        out.append((field, wire, v))
    return out


def read_events(path: str, verify_crc: bool = True) -> List[dict]:
    """Parse an event file into dicts: ``{wall_time, step, file_version?, graph?, scalars{}, histograms{}}``."""
    lib = native.host()
    events = []
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        hdr = data[pos:pos + 12]
        if len(hdr) < 12:
            raise ValueError("truncated record header")
        (length,), (crc_len,) = struct.unpack("<Q", hdr[:8]), struct.unpack("<I", hdr[8:])
        payload = data[pos + 12:pos + 12 + length]
        (crc_data,) = struct.unpack("<I", data[pos + 12 + length:pos + 16 + length])
        if verify_crc:
            if lib.dtf_masked_crc32c(hdr[:8], 8) != crc_len:
                raise ValueError("length CRC mismatch at %d" % pos)
            if lib.dtf_masked_crc32c(payload, len(payload)) != crc_data:
                raise ValueError("data CRC mismatch at %d" % pos)
        pos += 16 + length
        ev = {"scalars": {}, "histograms": {}}
        for field, wire, v in _parse_fields(payload):
            if field == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif field == 2:
                ev["step"] = v
            elif field == 3:
                ev["file_version"] = v.decode()
            elif field == 4:
                ev["graph"] = v
            elif field == 5:
                for f2, _w2, val in _parse_fields(v):
                    if f2 != 1:
                        continue
                    tag, simple, histo = None, None, None
                    for f3, _w3, x in _parse_fields(val):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2:
                            simple = struct.unpack("<f", x)[0]
                        elif f3 == 5:
                            histo = {}
                            names = {1: "min", 2: "max", 3: "num", 4: "sum", 5: "sum_squares"}
                            for f4, _w4, y in _parse_fields(x):
                                if f4 in names:
                                    histo[names[f4]] = struct.unpack("<d", y)[0]
                                elif f4 == 6:
                                    histo["bucket_limit"] = list(struct.unpack("<%dd" % (len(y) // 8), y))
                                elif f4 == 7:
                                    histo["bucket"] = list(struct.unpack("<%dd" % (len(y) // 8), y))
                    if simple is not None:
                        ev["scalars"][tag] = simple
                    if histo is not None:
                        ev["histograms"][tag] = histo
        events.append(ev)
    return events
