#!/bin/bash
# GPU box: does it matter on which socket the CLI's threads run?  Topology, then the CLI on the same clips pinned to the CPUs of each NUMA
# node (taskset), left to the scheduler, and with the CLI's own binding to the node of the device; interleaved.  usage: numa_probe.sh [rounds]
rounds=${1:-3}
cd "$(dirname "$0")/.."
lscpu | grep -E "NUMA|Socket|Model name"
for d in /sys/class/drm/card*/device; do echo "$d numa_node $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null)"; done
for k in /sys/class/kfd/kfd/topology/nodes/*; do echo "$k: $(grep -E 'cpu_cores_count|simd_count' $k/properties | tr '\n' ' ')"; done 2>/dev/null | head -12
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tm_pkg import tm
w, h, bits, frames = 1920, 1080, 8, 1536
pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(4)]
for side, s in enumerate(("ref", "dis")):
    with open(f"/dev/shm/tm_numa_{s}.y4m", "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 C420jpeg\n".encode())
        blobs = [b"FRAME\n" + b"".join(pl.astype(np.uint8).tobytes() for pl in pr[side]) for pr in pairs]
        for i in range(frames):
            f.write(blobs[i % 4])
PY
run() { "$@" turbo-metrics_amd/bin/turbo-metrics /dev/shm/tm_numa_ref.y4m /dev/shm/tm_numa_dis.y4m -m ssimulacra2 --output json-lines 2>&1 >/dev/null | grep Processed | sed 's/.*frame pairs in //'; }
run env > /dev/null
nodes=$(ls -d /sys/devices/system/node/node* | wc -l)
for r in $(seq "$rounds"); do
  printf 'default (binds to the node of the device) | '; run env
  printf 'TM_NUMA_BIND=0 (left to the scheduler)    | '; run env TM_NUMA_BIND=0
  for n in $(seq 0 $((nodes - 1))); do
    cpus=$(cat /sys/devices/system/node/node$n/cpulist)
    printf 'taskset node %d (%s), TM_NUMA_BIND=0 | ' "$n" "$cpus"; run env TM_NUMA_BIND=0 taskset -c "$cpus"
  done
done
rm -f /dev/shm/tm_numa_ref.y4m /dev/shm/tm_numa_dis.y4m
