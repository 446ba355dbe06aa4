#!/bin/bash
# GPU box: A/B of CLI variants on the same clips, interleaved.  usage: cli_ab.sh <1080p|4k> <rounds> "<args A>" "<args B>" ...
size=$1; rounds=$2; shift; shift
cd "$(dirname "$0")/.."
python3 - "$size" <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tm_pkg import tm
size = sys.argv[1]
w, h, bits, frames = (1920, 1080, 8, 1536) if size == "1080p" else (3840, 2160, 10, 256)
pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(4)]
for side, s in enumerate(("ref", "dis")):
    with open(f"/dev/shm/tm_cli_{size}_{s}.y4m", "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 C420{'jpeg' if bits == 8 else 'p10'}\n".encode())
        blobs = [b"FRAME\n" + b"".join(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes() for pl in pr[side]) for pr in pairs]
        for i in range(frames):
            f.write(blobs[i % 4])
PY
# a variant's words of the form TM_...=value (GPU_...=, HIP_...=) go into the environment, the others onto the command line; a variant that uses --tune runs with
# the laboratory build of the engine library in front of the ship library the CLI links (tuning values are laboratory functions)
run() { local envs=() args=(); for t in $1; do if [[ $t == TM_*=* || $t == GPU_*=* || $t == HIP_*=* ]]; then envs+=("$t"); else args+=("$t"); fi; done; if [[ $1 == *--tune* ]]; then envs+=("LD_PRELOAD=$PWD/turbo-metrics_amd/lab/libturbometrics_hip_lab.so"); fi; set -- "${args[*]}"; env RUST_LOG=debug "${envs[@]}" turbo-metrics_amd/bin/turbo-metrics /dev/shm/tm_cli_${size}_ref.y4m /dev/shm/tm_cli_${size}_dis.y4m -m ssimulacra2 --output json-lines $1 2>&1 >/dev/null | grep -E "Processed|main thread" | sed -e 's/.*Processed: [0-9]* (decoded: ~[0-9]*) frame pairs in //' -e 's/.*main thread: /   /' | tr '\n' ' '; echo; }
run "" > /dev/null   # first pass over the fresh files
for r in $(seq "$rounds"); do
  for v in "$@"; do printf '%-40s | ' "[$v]"; run "$v"; done
done
rm -f /dev/shm/tm_cli_${size}_ref.y4m /dev/shm/tm_cli_${size}_dis.y4m
