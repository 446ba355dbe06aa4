#!/bin/bash
# GPU box: where the CLI's main thread spends a run (RUST_LOG=debug prints the loop's own accounting).  usage: cli_timing.sh [1080p|4k] [runs] [extra CLI args]
size=${1:-1080p}; runs=${2:-4}; shift; shift
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
for i in $(seq "$runs"); do
  RUST_LOG=debug turbo-metrics_amd/bin/turbo-metrics /dev/shm/tm_cli_${size}_ref.y4m /dev/shm/tm_cli_${size}_dis.y4m -m ssimulacra2 --output json-lines "$@" 2>&1 >/dev/null | grep -E "Processed|main thread" | sed 's/.*turbo_metrics_cli: //'
done
rm -f /dev/shm/tm_cli_${size}_ref.y4m /dev/shm/tm_cli_${size}_dis.y4m
