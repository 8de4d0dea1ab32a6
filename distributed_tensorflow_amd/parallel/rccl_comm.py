"""Native RCCL communicator (``csrc/hip/rccl_comm.hip``; SURVEY §5.8, component 5.8 / T1-T4).

One ``ncclComm_t`` per process, created from a ``ncclUniqueId`` that rank 0 publishes through the framework's own
rendezvous store (``csrc/host/tcp_store.cc``) — the replacement for the in-process gRPC server every reference worker
starts (``/root/reference/demo2/train.py:13-21``).  The collectives are enqueued on the caller's HIP stream, so they can
be captured into the step's hipGraph: the RCCL data-parallel step is ONE graph launch.

``torch.distributed`` is used for one thing only, and only when no store address is given: rank 0 telling the others
which port its store listens on.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from .. import native
from .cluster import StoreClient, StoreServer


def _signatures(lib) -> None:
    if getattr(lib, "_dtf_rccl_sigs", False):
        return
    P, I, L = C.c_void_p, C.c_int, C.c_long
    lib.dtf_rccl_last_error.argtypes, lib.dtf_rccl_last_error.restype = [], C.c_char_p
    lib.dtf_rccl_load.argtypes, lib.dtf_rccl_load.restype = [C.c_char_p, C.POINTER(I)], I
    lib.dtf_rccl_unique_id.argtypes, lib.dtf_rccl_unique_id.restype = [P], I
    lib.dtf_rccl_init.argtypes, lib.dtf_rccl_init.restype = [P, I, I], P
    lib.dtf_rccl_allreduce_f32.argtypes, lib.dtf_rccl_allreduce_f32.restype = [P, P, P, L, I, P], I
    lib.dtf_rccl_allreduce_bf16.argtypes, lib.dtf_rccl_allreduce_bf16.restype = [P, P, L, P], I
    lib.dtf_rccl_allreduce_f32_multi.argtypes, lib.dtf_rccl_allreduce_f32_multi.restype = [P, P, P, I, P], I
    lib.dtf_rccl_broadcast.argtypes, lib.dtf_rccl_broadcast.restype = [P, P, L, I, P], I
    lib.dtf_rccl_allgather.argtypes, lib.dtf_rccl_allgather.restype = [P, P, P, L, P], I
    lib.dtf_rccl_async_error.argtypes, lib.dtf_rccl_async_error.restype = [P], I
    lib.dtf_rccl_destroy.argtypes, lib.dtf_rccl_destroy.restype = [P, I], None
    lib._dtf_rccl_sigs = True


def _torch_rccl_path() -> bytes:
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else b""


class RcclError(RuntimeError):
    pass


class RcclCommunicator:
    """``RcclCommunicator.create(rank, world, device)`` is collective over the ranks of the job."""

    def __init__(self, handle: int, rank: int, world: int, device, version: int, server: Optional[StoreServer]):
        self.lib = native.hip()
        self._h = handle
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        self.version = version
        self._server = server  # (rank 0 keeps the bootstrap store alive for the communicator's lifetime)

    # ------------------------------------------------------------------ bootstrap
    @staticmethod
    def available() -> bool:
        lib = native.hip()
        _signatures(lib)
        v = C.c_int(0)
        return lib.dtf_rccl_load(_torch_rccl_path(), C.byref(v)) == 0

    @classmethod
    def create(cls, rank: int, world: int, device, *, store: Optional[StoreClient] = None, key: str = "rccl/0",
               store_addr: Optional[str] = None, timeout_s: float = 60.0) -> "RcclCommunicator":
        """Rank 0 draws the unique id and publishes it under ``key``; everyone joins the communicator.
        ``store``: an open client of the job's rendezvous store; else ``store_addr`` ("host:port", rank 0 hosts it);
        else rank 0 hosts a store on an ephemeral port and announces the port through ``torch.distributed``."""
        lib = native.hip()
        _signatures(lib)
        ver = C.c_int(0)
        if lib.dtf_rccl_load(_torch_rccl_path(), C.byref(ver)) != 0:
            raise RcclError("RCCL unavailable: %s" % lib.dtf_rccl_last_error().decode())
        torch.cuda.set_device(torch.device(device))
        server = None
        own_client = None
        if store is None:
            import torch.distributed as dist
            host, port = None, 0
            if store_addr:
                host, p = store_addr.rsplit(":", 1)
                port = int(p)
            if rank == 0:
                server = StoreServer(port)
                port = server.port
            if host is None:
                if not dist.is_initialized():
                    raise RcclError("no rendezvous store address given and torch.distributed is not initialised")
                box = [port]
                dist.broadcast_object_list(box, src=0)
                port = int(box[0])
                host = os.environ.get("MASTER_ADDR", "127.0.0.1")
            store = own_client = StoreClient("127.0.0.1" if rank == 0 else host, port, timeout_s=timeout_s)
        try:
            if rank == 0:
                buf = (C.c_char * 128)()
                rc = lib.dtf_rccl_unique_id(buf)
                if rc:
                    raise RcclError("ncclGetUniqueId: %s" % lib.dtf_rccl_last_error().decode())
                store.set(key, bytes(buf))
                uid = bytes(buf)
            else:
                uid = store.get(key, timeout_s=timeout_s, cap=256)
                if uid is None or len(uid) != 128:
                    raise RcclError("rank %d: no RCCL unique id under %r within %.0f s" % (rank, key, timeout_s))
            h = lib.dtf_rccl_init(uid, rank, world)
            if not h:
                raise RcclError("ncclCommInitRank (rank %d of %d): %s" % (rank, world, lib.dtf_rccl_last_error().decode()))
            # everyone has read the id before rank 0 may tear the store down
            store.add(key + "/joined", 1)
            if rank == 0:
                store.wait_counter(key + "/joined", world, timeout_s=timeout_s)
        finally:
            if own_client is not None:
                own_client.close()
        return cls(h, rank, world, device, int(ver.value), server)

    # ------------------------------------------------------------------ collectives (enqueued on a stream; capturable)
    def _st(self, stream) -> int:
        if stream is None:
            return torch.cuda.current_stream(self.device).cuda_stream
        return stream if isinstance(stream, int) else stream.cuda_stream

    def _check(self, rc: int, what: str) -> None:
        if rc:
            raise RcclError("%s failed (%d): %s" % (what, rc, self.lib.dtf_rccl_last_error().decode()))

    def all_reduce(self, t: torch.Tensor, stream=None, op: str = "sum") -> None:
        """In-place fp32 sum (or max), or bf16 sum, over the ranks."""
        if t.dtype == torch.bfloat16:
            assert t.is_cuda and t.is_contiguous() and op == "sum"
            self._check(self.lib.dtf_rccl_allreduce_bf16(self._h, t.data_ptr(), t.numel(), self._st(stream)), "ncclAllReduce")
            return
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._check(self.lib.dtf_rccl_allreduce_f32(self._h, t.data_ptr(), t.data_ptr(), t.numel(), int(op == "max"),
                                                    self._st(stream)), "ncclAllReduce")

    def all_reduce_multi(self, tensors: Sequence[torch.Tensor], stream=None) -> None:
        """Several in-place fp32 sums as one RCCL group."""
        n = len(tensors)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        cnts = (C.c_long * n)(*[t.numel() for t in tensors])
        self._check(self.lib.dtf_rccl_allreduce_f32_multi(self._h, ptrs, cnts, n, self._st(stream)), "ncclAllReduce group")

    def broadcast(self, t: torch.Tensor, root: int = 0, stream=None) -> None:
        assert t.is_cuda and t.is_contiguous()
        self._check(self.lib.dtf_rccl_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), root, self._st(stream)),
                    "ncclBroadcast")

    def all_gather(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """``[world, *t.shape]``: every rank's ``t``."""
        assert t.is_cuda and t.is_contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._check(self.lib.dtf_rccl_allgather(self._h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(),
                                                self._st(stream)), "ncclAllGather")
        return out

    def async_error(self) -> Optional[str]:
        """None while the communicator is healthy, else the asynchronous RCCL error (a peer died, a transport failed)."""
        rc = self.lib.dtf_rccl_async_error(self._h)
        return None if rc == 0 else self.lib.dtf_rccl_last_error().decode()

    def close(self, abort: bool = False) -> None:
        if self._h:
            torch.cuda.synchronize(self.device)
            self.lib.dtf_rccl_destroy(self._h, int(abort))
            self._h = None
        if self._server is not None:
            self._server.stop()
            self._server = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
