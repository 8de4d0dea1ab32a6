"""Single-node launcher: spawn N worker ranks (one per GPU) with localhost cluster flags.

The reference expects three processes started by hand on three LAN hosts
(``demo2/train.py:201,207``); ``--num_gpus N`` (or this helper) starts the same script N times with
``--job_name=worker --task_index=r --worker_hosts=127.0.0.1:p0,...`` plus the torchrun-style
environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Children are started
before anything in this process touches the GPU, and the launcher exits with the worst child code.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_local_workers(script: str, argv: List[str], nproc: int, timeout_s: float = None) -> int:
    port = free_port()
    hosts = ",".join("127.0.0.1:%d" % (port + i) for i in range(nproc))
    # drop cluster flags given by the caller; we provide consistent ones
    clean, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a.split("=")[0] in ("--worker_hosts", "--ps_hosts", "--job_name", "--task_index"):
            skip = "=" not in a
            continue
        clean.append(a)
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        cmd = [sys.executable, script] + clean + ["--job_name=worker", "--task_index=%d" % r,
                                                  "--worker_hosts=" + hosts, "--ps_hosts="]
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    for p in procs:
        try:
            rc = max(rc, abs(p.wait(timeout=timeout_s)))
        except subprocess.TimeoutExpired:
            p.kill()
            rc = max(rc, 124)
    return rc
