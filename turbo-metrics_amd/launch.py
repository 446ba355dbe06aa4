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
from typing import List, Optional, Set


def effective_cpus() -> int:
    """CPUs this process may really use: the smallest of the visible CPUs, the affinity mask and the cgroup CPU quota.  The GPU boxes
    of this pool show 256 CPUs and allow 16 CPUs' worth of time (cpu.max = "1600000 100000"): more busy threads than that only buy
    throttling (round 3's "all cores" figure: 256 threads delivered less than 64)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                q, per = open(quota_file).read().split()[:2]
            else:
                q, per = open(quota_file).read().strip(), open(period_file).read().strip()
            if q != "max" and int(q) > 0 and int(per) > 0:
                n = min(n, max(1, -(-int(q) // int(per))))
                break
        except Exception:
            continue
    return n


def cap_rank_threads(world: int) -> int:
    """FIRST thing in a rank process, before numpy / torch are imported: this rank's share of the CPUs the job may use.  N ranks each
    starting torch's default intra-op pool (one thread per VISIBLE CPU: 256 on this pool's boxes) on a 16-CPU quota spend their setup
    being throttled; OMP / MKL / OpenBLAS read these variables when they load.  A smaller value that is already set (torchrun exports
    OMP_NUM_THREADS=1 for nproc > 1) is kept.  Returns the share; the caller passes it to torch.set_num_threads and sizes its own
    thread pools with it."""
    share = max(1, effective_cpus() // max(1, world))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        try:
            have = int(os.environ.get(var, "0"))
        except ValueError:
            have = 0
        os.environ[var] = str(min(have, share) if have > 0 else share)
    return share


def parse_cpulist(text: str) -> Set[int]:
    """"0-63,128-191" -> {0..63, 128..191} (the format of /sys/devices/system/node/nodeN/cpulist); empty on anything else"""
    out: Set[int] = set()
    try:
        for part in text.strip().split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            lo_i, hi_i = int(lo), int(hi or lo)
            if lo_i < 0 or hi_i < lo_i:
                return set()
            out.update(range(lo_i, hi_i + 1))
    except ValueError:
        return set()
    return out


def pci_numa_node(domain: int, bus: int, device: int, sysfs: str = "/sys/bus/pci/devices") -> int:
    """NUMA node of a PCI device from sysfs (function 0), -1 when unknown.  The fallback for tm_device_numa_node under a HIP runtime
    that does not know hipDeviceAttributeHostNumaId (the one bundled with torch answers -1; /opt/rocm's, which the CLI loads, does)."""
    try:
        v = int(open(os.path.join(sysfs, f"{domain:04x}:{bus:02x}:{device:02x}.0", "numa_node")).read().strip())
        return v if v >= 0 else -1
    except (OSError, ValueError):
        return -1


def bind_to_numa_node(node: int, sysfs: str = "/sys/devices/system/node") -> Optional[int]:
    """Bind the calling thread (and every thread it starts afterwards) to the CPUs of host NUMA node `node` that the process may run
    on -- the rule of the CLI's bind_to_device_node (host/turbo_metrics.cpp): page-locked memory then lands next to the device and the
    copy engines do not cross the socket link (1.4-1.8 x on the host-fed paths, profiles/r04y_numa_probe.log).  `node` comes from
    tm_device_numa_node(local_rank) in the rank itself, AFTER tm_init: the launcher parent stays GPU-free.  TM_NUMA_BIND=0 turns it
    off.  Returns the number of CPUs bound to, or None when nothing was changed (unknown node, nothing to narrow, not permitted)."""
    if node is None or node < 0 or os.environ.get("TM_NUMA_BIND") == "0":
        return None
    try:
        want = parse_cpulist(open(os.path.join(sysfs, f"node{node}", "cpulist")).read())
        now = os.sched_getaffinity(0)
        both = want & now
        if not both or both == now:
            return None
        os.sched_setaffinity(0, both)
        return len(both)
    except (OSError, AttributeError):
        return None


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
