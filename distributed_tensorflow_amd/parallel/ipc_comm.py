"""Python side of the direct xGMI / HIP-IPC all-reduce (``csrc/hip/ipc_allreduce.hip``).

``IpcAllReduce`` allocates this rank's fine-grained exchange buffer, swaps IPC handles with the other ranks through
the already-initialised ``torch.distributed`` group (any backend: the handles are 64-byte blobs), maps the peers and
then reduces with three kernel launches on the caller's stream — no host round trip, capturable into the step's
hipGraph.  ``self_test`` compares a few reductions of random data against ``torch.distributed.all_reduce`` and
agrees on the verdict across ranks, so a box where peer mapping misbehaves falls back to RCCL instead of training on
wrong sums.  Needs ``HSA_ENABLE_IPC_MODE_LEGACY=0`` (dmabuf IPC) on this pool.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import native

MAX_RANKS = 8


class _IpcCommStruct(C.Structure):  # csrc/hip/ipc_allreduce.hip: struct IpcComm
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("n_max", C.c_long), ("base", C.c_void_p * MAX_RANKS),
                ("epoch", C.c_void_p), ("error", C.c_void_p), ("timeout_ticks", C.c_ulonglong), ("skip", C.c_void_p)]


def _signatures(lib) -> None:
    if getattr(lib, "_dtf_ipc_sigs", False):
        return
    P, I, L = C.c_void_p, C.c_int, C.c_long
    lib.dtf_ipc_alloc.argtypes, lib.dtf_ipc_alloc.restype = [L, C.POINTER(P)], I
    lib.dtf_ipc_free.argtypes, lib.dtf_ipc_free.restype = [P], I
    lib.dtf_ipc_get_handle.argtypes, lib.dtf_ipc_get_handle.restype = [P, P], I
    lib.dtf_ipc_open_handle.argtypes, lib.dtf_ipc_open_handle.restype = [P, C.POINTER(P)], I
    lib.dtf_ipc_close_handle.argtypes, lib.dtf_ipc_close_handle.restype = [P], I
    lib.dtf_ipc_allreduce.argtypes, lib.dtf_ipc_allreduce.restype = [P, P, P, L, I, P], I
    lib.dtf_ipc_allreduce_zc.argtypes, lib.dtf_ipc_allreduce_zc.restype = [P, P, L, L, L, I, P], I
    lib.dtf_ipc_sync.argtypes, lib.dtf_ipc_sync.restype = [P, P], I
    lib.dtf_ipc_reset_errors.argtypes, lib.dtf_ipc_reset_errors.restype = [P], I
    lib.dtf_ipc_all_gather.argtypes, lib.dtf_ipc_all_gather.restype = [P, P, P, L, I, P], I
    lib.dtf_ipc_broadcast.argtypes, lib.dtf_ipc_broadcast.restype = [P, P, L, I, I, P], I
    lib.dtf_ipc_allreduce_oneshot.argtypes, lib.dtf_ipc_allreduce_oneshot.restype = [P, P, P, L, I, P], I
    lib.dtf_ipc_publish.argtypes, lib.dtf_ipc_publish.restype = [P, P, L, I, P], I
    lib.dtf_ipc_stage_ptr.argtypes, lib.dtf_ipc_stage_ptr.restype = [P], P
    lib.dtf_ipc_result_ptr.argtypes, lib.dtf_ipc_result_ptr.restype = [P, L], P
    lib._dtf_ipc_sigs = True


class IpcAllReduce:
    def __init__(self, n_max: int, device, group=None, timeout_s: float = 2.0, blocks: int = 64, skip_ptr: int = 0):
        """``skip_ptr``: device address of the engine's [skip flag, skipped-step count] words: when a wait of an exchange
        expires on any rank, every rank that can know raises the flag and the optimizer launch does not apply the step
        (``csrc/hip/ipc_allreduce.hip``: settle_waits)."""
        if not dist.is_initialized():
            raise RuntimeError("IpcAllReduce needs an initialised torch.distributed group for the handle exchange")
        self.lib = native.hip()
        _signatures(self.lib)
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if self.world > MAX_RANKS:
            raise ValueError("the IPC all-reduce serves one node (<= %d ranks)" % MAX_RANKS)
        self.device = torch.device(device)
        self.n_max = (int(n_max) + 3) // 4 * 4
        self.blocks = blocks
        self._own = None
        self._mapped: List[int] = []
        self.comm = None
        # Every rank walks through the SAME sequence of collectives (handle all-gather, barrier) whether or not its
        # local HIP calls succeed: a rank that bailed out early would leave its peers blocked in a collective.
        self.failure: Optional[str] = None
        hbytes = b"\0" * 64
        try:
            torch.cuda.set_device(self.device)
            own = C.c_void_p()
            native.check(self.lib.dtf_ipc_alloc(self.n_max, C.byref(own)), "ipc_alloc")
            self._own = own.value
            hbuf = (C.c_char * 64)()
            native.check(0 if self.lib.dtf_ipc_get_handle(self._own, hbuf) > 0 else -1, "hipIpcGetMemHandle")
            hbytes = bytes(hbuf)
        except Exception as e:  # noqa: BLE001
            self.failure = repr(e)
        handles: List[Optional[tuple]] = [None] * self.world
        dist.all_gather_object(handles, (hbytes, self.failure is None), group=group)
        if self.failure is None and not all(ok for _h, ok in handles):
            self.failure = "a peer could not allocate/export its exchange buffer"
        if self.failure is None:
            try:
                bases = (C.c_void_p * MAX_RANKS)()
                for p in range(self.world):
                    if p == self.rank:
                        bases[p] = self._own
                        continue
                    ptr = C.c_void_p()
                    native.check(self.lib.dtf_ipc_open_handle(C.create_string_buffer(handles[p][0], 64), C.byref(ptr)),
                                 "hipIpcOpenMemHandle (rank %d -> %d)" % (self.rank, p))
                    bases[p] = ptr.value
                    self._mapped.append(ptr.value)
                self._bases = [int(bases[p]) for p in range(self.world)]
                # epoch, error, arrival ticket, error total acted on, decision tag, spare
                self._words = torch.zeros(6, dtype=torch.int32, device=self.device)
                self.comm = _IpcCommStruct(self.rank, self.world, self.n_max, bases, self._words.data_ptr(),
                                           self._words.data_ptr() + 4, int(timeout_s * 100e6), skip_ptr or None)
                torch.cuda.synchronize(self.device)
            except Exception as e:  # noqa: BLE001
                self.failure = repr(e)
        dist.barrier(group=group)  # every rank has mapped every peer before the first flag is written

    def all_reduce(self, src: torch.Tensor, dst: Optional[torch.Tensor] = None, stream: Optional[int] = None,
                   blocks: Optional[int] = None) -> None:
        """dst <- sum over ranks of src (fp32, numel % 4 == 0, <= n_max); in place when ``dst`` is None."""
        dst = src if dst is None else dst
        if self.comm is None:
            raise RuntimeError("IPC all-reduce is not usable on this rank: %s" % self.failure)
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
        assert src.numel() == dst.numel() and src.numel() % 4 == 0 and src.numel() <= self.n_max
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        native.check(self.lib.dtf_ipc_allreduce(C.byref(self.comm), src.data_ptr(), dst.data_ptr(), src.numel(),
                                                self.blocks if blocks is None else blocks, st),
                     "ipc_allreduce")

    @property
    def stage_ptr(self) -> int:
        """Device address of this rank's stage buffer (n_max floats): peers read it during the reduction."""
        return int(self.lib.dtf_ipc_stage_ptr(self._own))

    @property
    def result_ptr(self) -> int:
        """Device address of this rank's result buffer (n_max floats): peers write their reduced slices into it."""
        return int(self.lib.dtf_ipc_result_ptr(self._own, self.n_max))

    def all_reduce_oneshot(self, src: torch.Tensor, dst: Optional[torch.Tensor] = None, stream: Optional[int] = None,
                           blocks: int = 64) -> None:
        """Small-buffer all-reduce in two kernels (every rank sums all ranks' buffers itself).  Contract: between two
        calls every rank passes another barrier over the same ranks (see ``dtf_ipc_allreduce_oneshot``)."""
        dst = src if dst is None else dst
        if self.comm is None:
            raise RuntimeError("IPC all-reduce is not usable on this rank: %s" % self.failure)
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
        assert src.numel() == dst.numel() and src.numel() % 4 == 0 and src.numel() <= self.n_max
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        native.check(self.lib.dtf_ipc_allreduce_oneshot(C.byref(self.comm), src.data_ptr(), dst.data_ptr(), src.numel(),
                                                        blocks, st), "ipc_allreduce_oneshot")

    def publish(self, src: torch.Tensor, stream: Optional[int] = None, blocks: int = 64) -> None:
        """First half of ``all_reduce_oneshot``: copy ``src`` into this rank's exchange buffer.  The consumer does the flag
        exchange and the reduction itself (``dtf_ipc_adam_oneshot``: the optimizer launch sums the ranks' buffers in its
        own gradient loads)."""
        if self.comm is None:
            raise RuntimeError("IPC all-reduce is not usable on this rank: %s" % self.failure)
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()
        assert src.numel() % 4 == 0 and src.numel() <= self.n_max
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        native.check(self.lib.dtf_ipc_publish(C.byref(self.comm), src.data_ptr(), src.numel(), blocks, st), "ipc_publish")

    # ---- the rest of a small collective backend over the same buffers; not on the training hot path
    def _stream(self, stream: Optional[int]) -> int:
        if self.comm is None:
            raise RuntimeError("IPC exchange is not usable on this rank: %s" % self.failure)
        return torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream

    def barrier(self, stream: Optional[int] = None) -> None:
        """Stream-ordered barrier over the ranks (one one-wave kernel; the host does not wait)."""
        native.check(self.lib.dtf_ipc_sync(C.byref(self.comm), self._stream(stream)), "ipc_sync")

    def all_gather(self, src: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        """``[world, n]`` <- every rank's ``src`` (fp32, numel % 4 == 0, <= n_max)."""
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.numel() % 4 == 0
        assert src.numel() <= self.n_max
        out = torch.empty(self.world, src.numel(), dtype=torch.float32, device=self.device)
        native.check(self.lib.dtf_ipc_all_gather(C.byref(self.comm), src.data_ptr(), out.data_ptr(), src.numel(), 64,
                                                 self._stream(stream)), "ipc_all_gather")
        return out

    def broadcast(self, t: torch.Tensor, src: int = 0, stream: Optional[int] = None) -> None:
        """``t`` (fp32, numel % 4 == 0, <= n_max) on every rank <- rank ``src``'s ``t``."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 and t.numel() <= self.n_max
        native.check(self.lib.dtf_ipc_broadcast(C.byref(self.comm), t.data_ptr(), t.numel(), src, 64, self._stream(stream)),
                     "ipc_broadcast")

    @property
    def comm_addr(self) -> int:
        """Host address of the communicator struct (kept alive by this object), for native callers."""
        return C.addressof(self.comm)

    def peer_stage_ptr(self, p: int) -> int:
        """Device address, in THIS process, of rank ``p``'s stage buffer."""
        return int(self.lib.dtf_ipc_stage_ptr(self._bases[p]))

    def all_reduce_zero_copy(self, src_rest: torch.Tensor, rest_off: int, n: int, stream: Optional[int] = None,
                             blocks: Optional[int] = None) -> None:
        """Sum over ranks of the ``n`` floats whose first ``rest_off`` already sit in this rank's stage buffer (written
        there by their producer) and whose tail is ``src_rest``; the sum is left in the result buffer (``result_ptr``)
        for the consumer to read in place.  Between this call and the next one nothing may write the stage buffer
        except that producer."""
        if self.comm is None:
            raise RuntimeError("IPC all-reduce is not usable on this rank: %s" % self.failure)
        assert src_rest.is_cuda and src_rest.dtype == torch.float32 and src_rest.is_contiguous()
        assert rest_off % 4 == 0 and src_rest.numel() % 4 == 0 and rest_off + src_rest.numel() == n <= self.n_max
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        native.check(self.lib.dtf_ipc_allreduce_zc(C.byref(self.comm), src_rest.data_ptr(), rest_off, src_rest.numel(), n,
                                                   self.blocks if blocks is None else blocks, st), "ipc_allreduce_zc")

    def set_timeout(self, timeout_s: float) -> None:
        """Bound of every in-kernel wait.  The struct is a by-value kernel argument: call this BEFORE the all-reduce is
        captured into a graph.  (Start-up uses a short bound so a broken mapping fails the self-test quickly; training
        wants a long one so a peer that pauses for an evaluation or a checkpoint is waited for, not given up on.)"""
        if self.comm is not None:
            self.comm.timeout_ticks = int(timeout_s * 100e6)

    def errors(self) -> int:
        """Number of waits that timed out so far.  Synchronises the DEVICE: the count is written by kernels on the engine's
        stream, and reading it from another stream while a step is still spinning in a wait returns the old count (seen
        with three ranks sharing a GPU: the reference schedule's first trial step ran into the 2 s bound after the count
        had been read, the step was skipped, and its zero "gradient" became the reference)."""
        torch.cuda.synchronize(self.device)
        return int(self._words[1].item())

    def reset_errors(self) -> None:
        """Forget the expired waits counted so far (synchronises).  For the end of the start-up trials: a candidate schedule
        whose waits expired was dropped and the training state restored, so its expiries must not make `ensure_healthy()`
        abandon the schedule that was chosen.  Clears this rank's count, the copy the peers read and the total this rank
        has acted on; every rank calls it, and the caller puts a barrier of the process group behind it (a rank that ran
        ahead would see a total different from the one it acted on, and skip a step)."""
        if self.comm is not None:
            native.check(self.lib.dtf_ipc_reset_errors(C.byref(self.comm)), "ipc_reset_errors")

    def self_test(self, n: int = 0, rounds: int = 3) -> bool:
        """Reduce random data ``rounds`` times and compare with the process group's own all-reduce; all ranks get the
        same verdict."""
        n = self.n_max if n <= 0 else min(self.n_max, (n + 3) // 4 * 4)
        gen = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        ok = True
        staged = dist.get_backend(self.group) == "gloo"
        # first agree that every rank mapped its peers: a kernel waiting for a rank that never signals would only
        # end by timeout
        usable = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int32)
        if not staged:
            usable = usable.to(self.device)
        dist.all_reduce(usable, op=dist.ReduceOp.MIN, group=self.group)
        if not int(usable.item()):
            return False
        for r in range(rounds):
            x = torch.randn(n, generator=gen).to(self.device)
            ref = x.cpu() if staged else x.clone()
            dist.all_reduce(ref, group=self.group)
            y = x.clone()
            self.all_reduce(y)
            torch.cuda.synchronize(self.device)
            ok = ok and bool(torch.allclose(y.cpu(), ref.cpu(), rtol=1e-5, atol=1e-5)) and self.errors() == 0
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if not staged:
            flag = flag.to(self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def close(self) -> None:
        """Collective: every rank of the group calls it, whether or not its own allocation / mapping succeeded (a rank
        that skipped the barrier would leave the others' next collective mismatched)."""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        torch.cuda.synchronize(self.device)
        if dist.is_initialized():
            dist.barrier(group=self.group)
        for p in self._mapped:
            self.lib.dtf_ipc_close_handle(p)
        self._mapped = []
        if getattr(self, "_own", None):
            self.lib.dtf_ipc_free(self._own)
            self._own = None
        self.comm = None
