"""Cluster spec, rendezvous and the (vestigial) parameter-server role (components C19, C20, C21).

Reference: ``tf.train.ClusterSpec({"ps": ps_hosts, "worker": worker_hosts})`` +
``tf.train.Server(cluster, job_name, task_index)``; ps tasks block in ``server.join()`` forever,
workers build their graph under ``replica_device_setter`` (``demo2/train.py:13-29``,
``retrain2/retrain2.py:369-380,411-413``).

MI355X mapping:

* parameters are replicated on every GPU, so there is no parameter server to talk to;
  ``--job_name=ps`` only hosts the native rendezvous store (``csrc/host/tcp_store.cc``) at its
  ``--ps_hosts`` address and exits cleanly once every worker reports done (quirk Q12: the
  reference's ps never terminates);
* ``len(worker_hosts)`` is the world size and ``--task_index`` the rank; rank -> GPU binding is
  ``local_rank = rank % visible GPUs`` (the replacement of ``replica_device_setter``);
* the collective backend is bootstrapped through torch.distributed's TCP rendezvous at the first
  worker's address (rank 0 == chief), RCCL for GPUs / gloo for CPU tests;
* the store carries job-level coordination: worker registration, heartbeats, the stop flag and
  the "done" counter.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import threading
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .. import native
from ..flags import parse_host_list


@dataclass
class ClusterSpec:
    jobs: Dict[str, List[str]] = field(default_factory=dict)

    @classmethod
    def from_flags(cls, ps_hosts: str, worker_hosts: str) -> "ClusterSpec":
        return cls({"ps": parse_host_list(ps_hosts), "worker": parse_host_list(worker_hosts)})

    def job_tasks(self, job: str) -> List[str]:
        return list(self.jobs.get(job, []))

    def num_tasks(self, job: str) -> int:
        return len(self.jobs.get(job, []))

    def task_address(self, job: str, index: int) -> str:
        return self.jobs[job][index]

    def as_dict(self) -> Dict[str, List[str]]:
        return {k: list(v) for k, v in self.jobs.items()}

    @property
    def world_size(self) -> int:
        return max(1, self.num_tasks("worker"))


def split_hostport(addr: str, default_port: int = 2222) -> Tuple[str, int]:
    if ":" in addr:
        h, p = addr.rsplit(":", 1)
        return h, int(p)
    return addr, default_port


class StoreServer:
    """In-process rendezvous store (what a ``ps`` task hosts)."""

    def __init__(self, port: int = 0):
        self._lib = native.host()
        self._h = self._lib.dtf_store_server_start(port)
        if not self._h:
            raise OSError("cannot start store server on port %d" % port)
        self.port = self._lib.dtf_store_server_port(self._h)

    def stop(self) -> None:
        if self._h:
            self._lib.dtf_store_server_stop(self._h)
            self._h = None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass


class StoreClient:
    def __init__(self, host: str, port: int, timeout_s: float = 30.0):
        self._lib = native.host()
        self._h = self._lib.dtf_store_connect(host.encode(), port, int(timeout_s * 1000))
        if not self._h:
            raise ConnectionError("cannot reach rendezvous store at %s:%d" % (host, port))
        # one connection, one request at a time: the heartbeat, monitor and training threads share this client
        self._lock = threading.Lock()

    def _handle(self):
        if not self._h:
            raise ConnectionError("store client is closed")
        return self._h

    def set(self, key: str, value: bytes) -> None:
        with self._lock:
            if self._lib.dtf_store_set(self._handle(), key.encode(), value, len(value)) < 0:
                raise ConnectionError("store set failed")

    def get(self, key: str, timeout_s: float = -1.0, cap: int = 1 << 16) -> Optional[bytes]:
        buf = C.create_string_buffer(cap)
        with self._lock:
            n = self._lib.dtf_store_get(self._handle(), key.encode(), buf, cap, int(timeout_s * 1000) if timeout_s >= 0 else -1)
        if n == -1:
            return None
        if n < 0:
            raise ConnectionError("store get failed (%d)" % n)
        return buf.raw[:min(n, cap)]

    def add(self, key: str, delta: int = 1) -> int:
        with self._lock:
            v = self._lib.dtf_store_add(self._handle(), key.encode(), delta)
        if v == -100:
            raise ConnectionError("store add failed")
        return v

    def check(self, key: str) -> bool:
        with self._lock:
            return self._lib.dtf_store_check(self._handle(), key.encode()) == 1

    def delete(self, key: str) -> None:
        with self._lock:
            self._lib.dtf_store_delete(self._handle(), key.encode())

    def counter(self, key: str) -> int:
        return self.add(key, 0)

    def wait_counter(self, key: str, target: int, timeout_s: float = 60.0, poll_s: float = 0.05) -> bool:
        t0 = time.time()
        while self.counter(key) < target:
            if time.time() - t0 > timeout_s:
                return False
            time.sleep(poll_s)
        return True

    def barrier(self, name: str, world: int, timeout_s: float = 60.0) -> bool:
        self.add("barrier/" + name, 1)
        return self.wait_counter("barrier/" + name, world, timeout_s)

    def close(self) -> None:
        with self._lock:
            if self._h:
                self._lib.dtf_store_close(self._h)
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Heartbeat:
    """Failure detection: every rank bumps ``hb/<rank>`` periodically; ``dead_ranks``
    reports ranks whose counter stopped moving (the reference has no heartbeat: a dead PS hangs
    the workers)."""

    def __init__(self, client: StoreClient, rank: int, period_s: float = 1.0):
        self.client, self.rank, self.period_s = client, rank, period_s
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._loop, daemon=True)
        self._last: Dict[int, Tuple[int, float]] = {}

    def start(self) -> "Heartbeat":
        self._th.start()
        return self

    def _loop(self) -> None:
        while not self._stop.wait(self.period_s):
            try:
                self.client.add("hb/%d" % self.rank, 1)
            except ConnectionError:
                return

    def dead_ranks(self, world: int, timeout_s: float) -> List[int]:
        now = time.time()
        dead = []
        for r in range(world):
            v = self.client.counter("hb/%d" % r)
            last = self._last.get(r)
            if last is None or v != last[0]:
                self._last[r] = (v, now)
            elif now - last[1] > timeout_s:
                dead.append(r)
        return dead

    def stop(self) -> None:
        self._stop.set()


def job_store(rank: int, world: int, timeout_s: float = 30.0):
    """The job's rendezvous store for ranks started by ``torchrun`` / the local launcher (no ``--ps_hosts``): rank 0 hosts
    one on an ephemeral port (or ``DTF_STORE_PORT``) and announces the port through the already initialised process group;
    everyone connects.  Returns ``(client, server or None)``; ``(None, None)`` for a single process."""
    if world <= 1:
        return None, None
    import torch.distributed as dist
    server = None
    port = int(os.environ.get("DTF_STORE_PORT", "0"))
    if rank == 0:
        server = StoreServer(port)
        port = server.port
    box = [port]
    dist.broadcast_object_list(box, src=0)
    host = "127.0.0.1" if rank == 0 else os.environ.get("MASTER_ADDR", "127.0.0.1")
    return StoreClient(host, int(box[0]), timeout_s=timeout_s), server


class Server:
    """``tf.train.Server(cluster, job_name, task_index)`` equivalent.

    ``ps``: ``join()`` serves the rendezvous store until every worker has reported done, then returns.
    ``worker``: ``target`` describes the collective rendezvous; ``init_collectives()`` joins it.
    """

    def __init__(self, cluster: ClusterSpec, job_name: str, task_index: int, *, backend: str = "auto",
                 start_store: bool = True):
        if job_name not in ("ps", "worker"):
            raise ValueError("job_name must be 'ps' or 'worker', got %r" % job_name)
        self.cluster, self.job_name, self.task_index, self.backend = cluster, job_name, task_index, backend
        self.store_server: Optional[StoreServer] = None
        self.store: Optional[StoreClient] = None
        ps = cluster.job_tasks("ps")
        self._store_addr = split_hostport(ps[0]) if ps else None
        if job_name == "ps" and task_index == 0 and start_store and self._store_addr:
            self.store_server = StoreServer(self._store_addr[1])

    @property
    def world_size(self) -> int:
        return self.cluster.world_size

    @property
    def rank(self) -> int:
        return self.task_index if self.job_name == "worker" else -1

    @property
    def is_chief(self) -> bool:
        return self.job_name == "worker" and self.task_index == 0

    @property
    def target(self) -> str:
        w = self.cluster.job_tasks("worker")
        return "tcp://%s" % (w[0] if w else "127.0.0.1:29511")

    def connect_store(self, timeout_s: float = 10.0) -> Optional[StoreClient]:
        """Connect to the ps-hosted store if one is reachable (optional: workers also run without a ps)."""
        if self.store is None and self._store_addr is not None:
            try:
                self.store = StoreClient(self._store_addr[0], self._store_addr[1], timeout_s)
            except ConnectionError:
                self.store = None
        return self.store

    def join(self, poll_s: float = 0.2, timeout_s: Optional[float] = None) -> None:
        """ps role: block until all workers signalled completion (or timeout), then stop the store."""
        if self.job_name != "ps":
            raise RuntimeError("join() is the parameter-server role")
        print("parameter server not needed: parameters are replicated on every GPU; "
              "hosting the rendezvous store on port %s" % (self.store_server.port if self.store_server else "-"))
        if self.store_server is None:
            return
        client = StoreClient("127.0.0.1", self.store_server.port, 5.0)
        t0 = time.time()
        try:
            while client.counter("workers_done") < self.world_size:
                if timeout_s is not None and time.time() - t0 > timeout_s:
                    break
                time.sleep(poll_s)
        finally:
            client.close()
            self.store_server.stop()

    def init_collectives(self, timeout_s: int = 600) -> Tuple[int, int]:
        """worker role: join the torch.distributed process group (RCCL on GPUs, gloo on CPU)."""
        from .ddp import init_process_group
        w = self.cluster.job_tasks("worker")
        host, port = split_hostport(w[0], 29511) if w else ("127.0.0.1", 29511)
        if self.world_size <= 1:
            return 0, 1
        return init_process_group(self.backend, rank=self.task_index, world=self.world_size,
                                  master_addr=host, master_port=port, timeout_s=timeout_s)

    def report_done(self) -> None:
        st = self.connect_store(timeout_s=1.0)
        if st is not None:
            try:
                st.add("workers_done", 1)
            except ConnectionError:
                pass


def bind_device(rank: int, device_flag: str = "auto"):
    """rank -> device (replacement of ``replica_device_setter``'s worker placement)."""
    import torch
    if device_flag == "cpu" or not torch.cuda.is_available():
        return torch.device("cpu")
    n = torch.cuda.device_count()
    # more local ranks than GPUs (functional runs on a small box): ranks wrap around and share devices
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, n)
    torch.cuda.set_device(local)
    return torch.device("cuda", local)
