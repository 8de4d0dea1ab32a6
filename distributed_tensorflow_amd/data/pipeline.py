"""Decode -> H2D -> resize/normalise -> backbone pipeline for bottleneck extraction (component C32/C33).

Reference: ``create_bottleneck_file`` / ``run_bottleneck_on_image`` (``/root/reference/retrain1/retrain.py:168-180,228-245``)
read, decode and push ONE image through Inception per ``sess.run``.  Round 1 batched the forward but decoded on a thread
pool under the GIL into a freshly pinned buffer per batch and serialised decode, copy and compute: 1.6-2.6 k images/s for a
forward that runs at 18 k images/s.

This pipeline keeps all three stages busy at once:

* decode: worker PROCESSES (``decode_worker.py``: numpy + PIL only), a chunk of a batch per task, written into a ring of
  shared-memory slots; a finished batch is moved with one ``memcpy`` (~1.5 ms for 15 MB) into a PERSISTENT ring of pinned
  buffers allocated once (no per-batch ``pin_memory()``).  (Page-locking the shared pages themselves would save that
  copy, but ``hipHostRegister`` returns "invalid argument" for every kind of host memory on this pool:
  ``scripts/dbg/host_register_probe.py``.)  JPEGs at least twice the input size are decoded at reduced scale (``draft``);
* H2D + resize/normalise: one DMA per batch (images are packed back to back at offsets known from a header probe) and
  one preprocessing launch, on a side stream, into one of two device input buffers;
* backbone: on the caller's stream, fenced by events against the side stream, so the copy of batch k+1 runs under the
  forward of batch k while the workers are decoding batches k+2 .. k+slots-1.

``stats`` reports where the time went (decode wait / H2D / GPU) for the log line the drivers print.
"""
from __future__ import annotations

import ctypes as C
import multiprocessing as mp
import os
import time
from concurrent.futures import ProcessPoolExecutor
from multiprocessing import shared_memory
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import native
from . import decode_worker as dw

INPUT_SIZE = 299


def default_workers() -> int:
    """CPU share of this process (cgroup / affinity aware), minus one for the feeding thread; at most 16."""
    env = os.environ.get("DTF_DECODE_WORKERS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 2
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):  # the harness publishes the box's CPU share through these
        if os.environ.get(var, "").isdigit():
            n = min(n, int(os.environ[var]))
    return max(1, min(16, n) - 1)


class DecodePipeline:
    def __init__(self, device, batch_size: int = 64, workers: Optional[int] = None, slots: int = 4, chunk: int = 8,
                 draft: int = 2 * INPUT_SIZE, max_image_bytes: int = 64 << 20, fp32: bool = False):
        self.device = torch.device(device)
        self.dtype = torch.float32 if fp32 else torch.bfloat16   # fp32: the input of the reference-precision forward
        self.batch_size, self.slots, self.chunk, self.draft = batch_size, max(2, slots), max(1, chunk), draft
        self.max_image_bytes = max_image_bytes
        self.workers = workers or default_workers()
        # spawn, not fork: the parent may already hold a HIP context, which a forked child must never inherit
        self.pool = ProcessPoolExecutor(max_workers=self.workers, mp_context=mp.get_context("spawn"))
        # A spawned child re-imports the parent's __main__ (a training script that imports torch: seconds and hundreds of
        # MB per worker).  The workers need nothing from it, so it is hidden while they are started (all of them start at
        # the first submit).
        import sys
        main = sys.modules.get("__main__")
        saved = {a: getattr(main, a) for a in ("__spec__", "__file__") if main is not None and hasattr(main, a)}
        try:
            if main is not None:
                main.__spec__ = None
                if "__file__" in saved:
                    del main.__file__
            self._warm = [self.pool.submit(dw.warm) for _ in range(self.workers)]
        finally:
            for a, v in saved.items():
                setattr(main, a, v)
        self._shm: List[shared_memory.SharedMemory] = []
        self._host: List[torch.Tensor] = []   # views of the shared slots (decode targets)
        self._pin: List[torch.Tensor] = []    # the pinned ring (DMA sources)
        self._slot_bytes = 0
        self.stats: Dict[str, float] = {}

    # ------------------------------------------------------------------ shared pinned ring
    def _ensure_ring(self, nbytes: int) -> None:
        if nbytes <= self._slot_bytes:
            return
        self._free_ring()
        nbytes = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
        for _ in range(self.slots):
            shm = shared_memory.SharedMemory(create=True, size=nbytes)
            self._shm.append(shm)
            self._host.append(torch.frombuffer(shm.buf, dtype=torch.uint8, count=nbytes))
            if self.device.type == "cuda":
                self._pin.append(torch.empty(nbytes, dtype=torch.uint8).pin_memory())
        self._slot_bytes = nbytes

    def _free_ring(self) -> None:
        if self.device.type == "cuda" and self._pin:
            torch.cuda.synchronize(self.device)
        self._pin = []
        self._host = []
        for shm in self._shm:
            try:
                shm.close()
            except Exception:  # (a view of the buffer is still alive somewhere: the mapping goes with the process)
                pass
            try:
                shm.unlink()
            except Exception:
                pass
        self._shm = []
        self._slot_bytes = 0

    def close(self) -> None:
        self.pool.shutdown(wait=True, cancel_futures=True)
        self._free_ring()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the pipeline
    def run(self, paths: Sequence[str], consume: Callable[[int, int, torch.Tensor], None],
            preprocess_cpu: Optional[Callable] = None, warm: Optional[Callable[[], None]] = None) -> Dict[str, float]:
        """Feed ``paths`` through decode -> H2D -> preprocess in batches of ``batch_size`` and call
        ``consume(first, count, x)`` for each, in order, on the current stream (see ``batches``).  Returns the stage
        statistics (also kept in ``self.stats``)."""
        for first, count, x in self.batches(paths, preprocess_cpu=preprocess_cpu, warm=warm):
            consume(first, count, x)
        return self.stats

    def batches(self, paths: Sequence[str], *, batch_size: Optional[int] = None, preprocess_cpu: Optional[Callable] = None,
                warm: Optional[Callable[[], None]] = None, desc_fn: Optional[Callable[[int], tuple]] = None,
                draft: Optional[int] = None):
        """Generator over ``(first, count, x)``: ``x`` is the NHWC ``[batch_size, 299, 299, 8]`` network input (``self.dtype``) of
        ``paths[first : first + count]`` (rows >= count repeat the last image), valid on the current stream until the
        next-but-one batch is requested.  ``warm()`` (one-time work of the consumer: kernel selection, graph capture) runs
        while the decoders fill the first slots.  ``desc_fn(i) -> (precrop, off_y, off_x, flip, brightness)`` gives image
        ``i`` a distortion (``bottleneck.Distortion.sample``); ``draft`` overrides the reduced-scale decode request (a
        distorted stream asks for its largest pre-crop size).  ``self.stats`` is filled when the stream is exhausted."""
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from .bottleneck import ImageDesc
        n, B = len(paths), int(batch_size or self.batch_size)
        draft = self.draft if draft is None else int(draft)
        self.stats = {}
        if n == 0:
            return
        t_all = time.perf_counter()
        for f in self._warm:
            f.result()
        self._warm = []
        # ---- header probe: decoded sizes -> packed offsets of every batch (parallel, header reads only, once per file)
        uniq = list(dict.fromkeys(paths))
        step = max(1, (len(uniq) + 4 * self.workers - 1) // (4 * self.workers))
        futs = [self.pool.submit(dw.probe, uniq[i:i + step], draft) for i in range(0, len(uniq), step)]
        size_of = dict(zip(uniq, (s for f in futs for s in f.result())))
        bad = [p for p, (h, w) in size_of.items() if h <= 0 or w <= 0 or h * w * 3 > self.max_image_bytes]
        if bad:
            raise ValueError("undecodable or oversized image(s): %s" % bad[:3])
        sizes = [size_of[p] for p in paths]
        t_probe = time.perf_counter() - t_all
        nb = (n + B - 1) // B
        offs = np.zeros(n, np.int64)
        batch_bytes = np.zeros(nb, np.int64)
        for k in range(nb):
            o = 0
            for i in range(k * B, min(n, (k + 1) * B)):
                offs[i] = o
                o += sizes[i][0] * sizes[i][1] * 3
            batch_bytes[k] = o
        self._ensure_ring(int(batch_bytes.max()))
        cuda = self.device.type == "cuda"
        if cuda:
            lib = native.hip()
            side = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            pix = [torch.empty(self._slot_bytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
            xbuf = [torch.empty(B, INPUT_SIZE, INPUT_SIZE, 8, dtype=self.dtype, device=self.device) for _ in range(2)]
            preprocess = lib.dtf_preprocess_images_f32 if self.dtype == torch.float32 else lib.dtf_preprocess_images
            dsz = C.sizeof(ImageDesc)
            desc_host = torch.empty(self.slots, B * dsz, dtype=torch.uint8).pin_memory()
            desc_dev = [torch.empty(B * dsz, dtype=torch.uint8, device=self.device) for _ in range(2)]
            x_ready = [torch.cuda.Event() for _ in range(2)]
            x_free = [torch.cuda.Event() for _ in range(2)]
            copied = [torch.cuda.Event() for _ in range(self.slots)]
            nev = min(nb, 64)
            t_ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(nev)]
            g_ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(2)) for _ in range(nev)]
            for e in x_free:
                e.record(main)

        def submit(k: int):
            s = k % self.slots
            lo, hi = k * B, min(n, (k + 1) * B)
            return [self.pool.submit(dw.decode_chunk, self._shm[s].name,
                                     [(paths[i], int(offs[i]), sizes[i][0], sizes[i][1]) for i in range(c0, min(hi, c0 + self.chunk))],
                                     draft)
                    for c0 in range(lo, hi, self.chunk)]

        pending = {k: submit(k) for k in range(min(nb, self.slots))}
        t_warm = time.perf_counter()
        if warm is not None:
            warm()
        t_warm = time.perf_counter() - t_warm

        def make_descs(k: int):
            lo, hi = k * B, min(n, (k + 1) * B)
            descs = (ImageDesc * B)()
            for j in range(B):
                i = min(lo + j, hi - 1)  # (a short last batch repeats its last image: one batch size, one graph)
                d = descs[j]
                H, W = sizes[i]
                precrop, oy, ox, flip, bright = desc_fn(i) if desc_fn is not None else (float(INPUT_SIZE), 0.0, 0.0, False, 1.0)
                d.offset, d.H, d.W = int(offs[i]), H, W
                d.scale_y, d.scale_x = H / float(precrop), W / float(precrop)
                d.off_y, d.off_x, d.flip, d.brightness = float(oy), float(ox), int(bool(flip)), float(bright)
            return descs

        # ---- feeder thread: waits for the decoders, moves the finished batch into its pinned buffer (the copy is split
        # over a few threads; it releases the GIL), hands the shared slot back to the decoders, and passes the batch on
        ready: "queue.Queue" = queue.Queue(maxsize=self.slots)
        tstat = {"wait": 0.0, "worker": 0.0, "copy": 0.0}
        # a pinned buffer may be refilled only after the consumer thread has ENQUEUED the DMA out of it (and recorded the
        # event the feeder then waits on): `issued` is the newest batch for which that has happened
        issued = [-1]
        issued_cv = threading.Condition()
        cancelled = [False]  # the consumer failed or abandoned the stream
        ncopy = 4
        copiers = ThreadPoolExecutor(max_workers=ncopy) if cuda else None

        def hand_over(item) -> bool:
            while not cancelled[0]:
                try:
                    ready.put(item, timeout=0.2)
                    return True
                except queue.Full:
                    pass
            return False

        def feeder() -> None:
            try:
                for k in range(nb):
                    t0 = time.perf_counter()
                    for f in pending.pop(k):
                        tstat["worker"] += f.result()[1]
                    tstat["wait"] += time.perf_counter() - t0
                    s = k % self.slots
                    nbytes = int(batch_bytes[k])
                    descs = make_descs(k)
                    imgs = None
                    if cuda:
                        t0 = time.perf_counter()
                        if k >= self.slots:
                            with issued_cv:
                                while issued[0] < k - self.slots and not cancelled[0]:
                                    issued_cv.wait(0.5)
                            if cancelled[0]:
                                return
                            copied[s].synchronize()  # the DMA out of this pinned buffer (batch k - slots) has finished
                        cut = [(nbytes * c // ncopy) & ~63 for c in range(ncopy)] + [nbytes]
                        list(copiers.map(lambda c: self._pin[s][cut[c]:cut[c + 1]].copy_(self._host[s][cut[c]:cut[c + 1]]),
                                         range(ncopy)))
                        C.memmove(desc_host[s].data_ptr(), descs, B * dsz)
                        tstat["copy"] += time.perf_counter() - t0
                    else:
                        lo, hi = k * B, min(n, (k + 1) * B)
                        imgs = []
                        for j in range(B):
                            i = min(lo + j, hi - 1)
                            H, W = sizes[i]
                            imgs.append(np.frombuffer(self._shm[s].buf, np.uint8, H * W * 3, int(offs[i])).reshape(H, W, 3).copy())
                    if k + self.slots < nb:
                        pending[k + self.slots] = submit(k + self.slots)  # the shared slot is free again
                    if not hand_over((k, descs, imgs)):
                        return
            except BaseException as e:  # noqa: BLE001 - hand the failure to the consuming thread
                hand_over(e)

        th = threading.Thread(target=feeder, name="dtf-extract-feeder", daemon=True)
        th.start()
        idle_s = 0.0
        completed = False
        try:
            for _ in range(nb):
                t0 = time.perf_counter()
                item = ready.get()
                idle_s += time.perf_counter() - t0
                if isinstance(item, BaseException):
                    raise item
                k, descs, imgs = item
                s, xb = k % self.slots, k % 2
                lo, hi = k * B, min(n, (k + 1) * B)
                nbytes = int(batch_bytes[k])
                if not cuda:
                    yield lo, hi - lo, preprocess_cpu(imgs, descs)
                    continue
                te = t_ev[k] if k < nev else None
                with torch.cuda.stream(side):
                    side.wait_event(x_free[xb])  # the consumer of batch k-2 is done with this input buffer
                    if te:
                        te[0].record(side)
                    pix[xb][:nbytes].copy_(self._pin[s][:nbytes], non_blocking=True)
                    desc_dev[xb].copy_(desc_host[s], non_blocking=True)
                    copied[s].record(side)
                    if te:
                        te[1].record(side)
                    native.check(preprocess(pix[xb].data_ptr(), desc_dev[xb].data_ptr(), xbuf[xb].data_ptr(), B, INPUT_SIZE,
                                            side.cuda_stream), "preprocess")
                    if te:
                        te[2].record(side)
                    x_ready[xb].record(side)
                with issued_cv:
                    issued[0] = k
                    issued_cv.notify_all()
                main.wait_event(x_ready[xb])
                if te:
                    g_ev[k][0].record(main)
                yield lo, hi - lo, xbuf[xb]
                if te:
                    g_ev[k][1].record(main)
                x_free[xb].record(main)
            completed = True
        finally:
            if not completed:  # the consumer failed or dropped the stream: let the feeder leave its waits
                cancelled[0] = True
                with issued_cv:
                    issued_cv.notify_all()
            th.join(timeout=30)
            if copiers is not None:
                copiers.shutdown()
            if cuda:
                torch.cuda.synchronize(self.device)
            # statistics of what was delivered (a consumer that stops after the last batch never resumes the generator)
            done_b = max(0, issued[0] + 1) if cuda else nb
            done_n = min(n, done_b * B)
            total = time.perf_counter() - t_all
            st = {"images": done_n, "seconds": total, "images_per_sec": done_n / max(total, 1e-9), "probe_s": t_probe,
                  "decode_wait_s": tstat["wait"], "decode_cpu_s": tstat["worker"], "host_copy_s": tstat["copy"],
                  "warm_s": t_warm, "consumer_idle_s": idle_s, "workers": self.workers, "batches": done_b,
                  "h2d_bytes": int(batch_bytes[:done_b].sum())}
            m = min(nev, done_b - 1) if cuda else 0  # (the last batch's closing event may never have been recorded)
            if m > 0:
                st["h2d_ms_per_batch"] = sum(a.elapsed_time(b) for a, b, _ in t_ev[:m]) / m
                st["preprocess_ms_per_batch"] = sum(b.elapsed_time(c) for _, b, c in t_ev[:m]) / m
                st["gpu_ms_per_batch"] = sum(a.elapsed_time(b) for a, b in g_ev[:m]) / m
            self.stats = st

    @staticmethod
    def describe(st: Dict[str, float]) -> str:
        if not st:
            return "nothing to extract"
        s = "%d images in %.2f s = %.0f images/s (%d decode processes)" % (st["images"], st["seconds"], st["images_per_sec"],
                                                                            st["workers"])
        s += ("; header probe %.2f s, one-time kernel selection + graph capture %.2f s; feeder thread: waiting for decoders "
              "%.2f s (%.2f s of decoder CPU time), shared->pinned copies %.2f s; GPU-feeding thread idle %.2f s" % (
                  st["probe_s"], st["warm_s"], st["decode_wait_s"], st["decode_cpu_s"], st["host_copy_s"], st["consumer_idle_s"]))
        if "gpu_ms_per_batch" in st:
            s += "; per batch: H2D %.2f ms (%.1f MB), resize/normalise %.2f ms, backbone %.2f ms" % (
                st["h2d_ms_per_batch"], st["h2d_bytes"] / st["batches"] / 1e6, st["preprocess_ms_per_batch"],
                st["gpu_ms_per_batch"])
        return s
