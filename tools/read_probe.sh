#!/bin/bash
# GPU box: the CLI's reader alone (file in tmpfs -> ring of page-locked surfaces, no GPU work): pictures/s and GB/s for 1, 2, 4, 7, 12
# reader threads, one stream and two streams at once; NUMA layout of the host.  usage: read_probe.sh TAG
set -u
TAG=${1:-rp}
R=$GRAFT_REPO_ROOT
H=$R/tests/host/tm_host_test
(lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)"; cat /sys/fs/cgroup/cpu.max; for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist) $(grep MemFree $n/meminfo)"; done; for d in /sys/class/drm/card*/device; do echo "$d numa_node $(cat $d/numa_node 2>/dev/null)"; done; cat /proc/self/status | grep -i "cpus_allowed_list\|mems_allowed_list") 2>&1 | tee gpurun_out/${TAG}_host.log
python3 - <<PY
import numpy as np, os
for tag, w, h, bits, frames in (("1080p", 1920, 1080, 8, 1024), ("4k", 3840, 2160, 10, 160)):
    per = w * h * 3 // 2 * (1 if bits == 8 else 2)
    rng = np.random.default_rng(1)
    blob = b"FRAME\n" + rng.integers(0, 256, per, dtype=np.uint8).tobytes()
    for s in ("a", "b"):
        with open(f"/dev/shm/rp_{tag}_{s}.y4m", "wb") as f:
            f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 C420{'jpeg' if bits == 8 else 'p10'}\n".encode())
            for i in range(frames):
                f.write(blob)
PY
for tag in 1080p 4k; do
  for t in 1 2 4 7 12 16; do
    echo "== $tag, $t reader threads, one stream"; TM_READER_THREADS=$t $H readbench /dev/shm/rp_${tag}_a.y4m 4
    echo "== $tag, $t reader threads per stream, two streams at once"
    TM_READER_THREADS=$t $H readbench /dev/shm/rp_${tag}_a.y4m 4 & TM_READER_THREADS=$t $H readbench /dev/shm/rp_${tag}_b.y4m 4; wait
  done
done 2>&1 | tee gpurun_out/${TAG}_read_probe.log
rm -f /dev/shm/rp_*.y4m
