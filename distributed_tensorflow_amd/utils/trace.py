"""Tracing / profiling helpers.

The reference has no tracing at all: its only timing is ``time.time()`` around the loop
(``demo1/train.py:152,164``) plus the Supervisor's ``global_step/sec`` scalar.  Here:

* ``range(name)`` / ``mark(name)`` — roctx ranges (``libroctx64``) that show up in ``rocprofv3 --marker-trace``
  timelines around the step phases (graph launches, bucket all-reduces, optimizer); no-ops unless ``enable()``d
  (``--profile``), so the hot loop pays one attribute test;
* ``StepTimer`` — host-side phase breakdown with optional device timing through HIP events, prints
  images/sec (the BASELINE metric) and the per-phase share;
* ``rocprof_stats(cmd)`` — run a command under ``rocprofv3 --kernel-trace --stats`` as a child process and return
  the per-kernel table (``summarize_db`` reads the sqlite file ROCm 7.2 writes).  Counter (``--pmc``) collection is
  deliberately a separate call: on this pool PMC must not be combined with the trace domains.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import sqlite3
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

_enabled = False
_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                lib = ctypes.CDLL(name)
                lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib.roctxRangePushA.restype = ctypes.c_int
                lib.roctxRangePop.restype = ctypes.c_int
                lib.roctxMarkA.argtypes = [ctypes.c_char_p]
                _roctx = lib
                break
            except (OSError, AttributeError):
                continue
        else:
            _roctx = False
    return _roctx


def enable(on: bool = True) -> bool:
    """Turn range emission on/off; returns whether roctx is actually available."""
    global _enabled
    _enabled = bool(on)
    return bool(_load_roctx()) if on else False


def enabled() -> bool:
    return _enabled


def push(name: str) -> None:
    if _enabled and _load_roctx():
        _roctx.roctxRangePushA(name.encode())


def pop() -> None:
    if _enabled and _load_roctx():
        _roctx.roctxRangePop()


def mark(name: str) -> None:
    if _enabled and _load_roctx():
        _roctx.roctxMarkA(name.encode())


@contextlib.contextmanager
def range(name: str):  # noqa: A001 - mirrors the roctx/nvtx naming
    if not _enabled:
        yield
        return
    push(name)
    try:
        yield
    finally:
        pop()


class StepTimer:
    """Accumulates wall time per named phase; ``device=True`` brackets phases with HIP events instead (resolved lazily
    so the hot loop never synchronises)."""

    def __init__(self, device: bool = False):
        self.device = device
        self.totals: Dict[str, float] = {}
        self.counts: Dict[str, int] = {}
        self._pending: List[Tuple[str, object, object]] = []
        self.t0 = time.perf_counter()
        self.images = 0

    @contextlib.contextmanager
    def phase(self, name: str):
        push(name)
        if self.device:
            import torch
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            try:
                yield
            finally:
                b.record()
                self._pending.append((name, a, b))
                pop()
        else:
            t = time.perf_counter()
            try:
                yield
            finally:
                self._add(name, time.perf_counter() - t)
                pop()

    def _add(self, name: str, dt: float) -> None:
        self.totals[name] = self.totals.get(name, 0.0) + dt
        self.counts[name] = self.counts.get(name, 0) + 1

    def add_images(self, n: int) -> None:
        self.images += n

    def resolve(self) -> None:
        if self._pending:
            import torch
            torch.cuda.synchronize()
            for name, a, b in self._pending:
                self._add(name, a.elapsed_time(b) / 1e3)
            self._pending.clear()

    def images_per_sec(self) -> float:
        return self.images / max(1e-9, time.perf_counter() - self.t0)

    def summary(self) -> str:
        self.resolve()
        tot = sum(self.totals.values()) or 1.0
        parts = ["%s %.3f ms x%d (%.0f%%)" % (k, 1e3 * v / max(1, self.counts[k]), self.counts[k], 100 * v / tot)
                 for k, v in sorted(self.totals.items(), key=lambda kv: -kv[1])]
        return "%.1f images/sec | " % self.images_per_sec() + " | ".join(parts)


# ------------------------------------------------------------------------------------------- rocprofv3
def summarize_db(path: str, top: int = 25) -> List[Tuple[str, int, float, float]]:
    """(kernel name, calls, total ms, avg us) rows of a rocprofv3 sqlite result, by total time."""
    db = sqlite3.connect(path)
    try:
        rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    finally:
        db.close()
    return [(n, int(c), t / 1e6, a / 1e3) for n, c, t, a in rows[:top]]


def format_table(rows: Sequence[Tuple[str, int, float, float]]) -> str:
    tot = sum(r[2] for r in rows) or 1.0
    lines = ["%-100s %7s %10s %6s %10s" % ("kernel", "calls", "total_ms", "%", "avg_us")]
    for n, c, t, a in rows:
        lines.append("%-100s %7d %10.2f %6.1f %10.1f" % (n[:100], c, t, 100 * t / tot, a))
    return "\n".join(lines)


def rocprof_stats(cmd: Sequence[str], outdir: str, name: str = "trace", timeout: Optional[float] = 600) -> str:
    """Run ``cmd`` (argv; the program itself, never a shell or ``env`` hop: the profiler initialises the GPU before the
    program starts) under ``rocprofv3 --kernel-trace --stats`` and return the formatted kernel table."""
    os.makedirs(outdir, exist_ok=True)
    full = ["rocprofv3", "--kernel-trace", "--stats", "-d", outdir, "-o", name, "--"] + list(cmd)
    subprocess.run(full, check=True, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return format_table(summarize_db(os.path.join(outdir, name + "_results.db")))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1].endswith(".db"):
        print(format_table(summarize_db(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)))
    else:
        print("usage: python -m distributed_tensorflow_amd.utils.trace <rocprofv3_results.db> [top_n]")
