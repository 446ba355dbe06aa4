"""One process per GPU on ONE node: `python bench.py --gpus N` without torchrun.

The parent that calls spawn_ranks() must not have touched the GPU (no HIP call, no torch.cuda.*): it only
starts N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- the same
environment `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would give them --, forwards rank 0's
stdout (the ONE JSON line), and returns non-zero when any rank fails.  The reference has no counterpart: it
is single-GPU (device 0 hard-coded, crates/turbo-metrics/src/lib.rs:442); SURVEY.md 8e defines the sharding.
"""
import os
import socket
import subprocess
import sys
import time
from typing import List, Optional


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(argv: List[str], n: int, timeout_s: Optional[float] = None, env_extra: Optional[dict] = None) -> int:
    """Run `argv` as n rank processes; rank 0's stdout goes to ours, every rank's stderr goes to ours (prefixed
    output would break nothing but is not needed).  Returns the first non-zero exit code, or 0.  When one rank
    dies the others are terminated (exact PIDs), so a failed rendezvous cannot hang the caller."""
    if n < 1:
        raise ValueError("n >= 1")
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.monotonic()
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        failed = rc != 0 or (timeout_s is not None and time.monotonic() - t0 > timeout_s)
        if failed and alive:
            if rc == 0:
                rc = 124
            for p in alive:
                p.terminate()
            for p in alive:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            alive = []
        if alive:
            time.sleep(0.05)
    if rc != 0:
        print(f"launch: a rank failed (exit code {rc}); {n} ranks were started", file=sys.stderr)
    return rc
