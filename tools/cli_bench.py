#!/usr/bin/env python3
"""GPU box: end-to-end throughput of the `turbo-metrics` CLI on a host-fed Y4M stream (read + repack + upload + metric).
usage: cli_bench.py [--frames N] [--size 1080p|4k] -- prints the CLI's own "Processed ... fps" line per configuration."""
import argparse, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tm_pkg import tm

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=0, help="pairs per clip (0 = 1536 at 1080p, 256 at 4K: long enough that ring page-locking and first-touch of the mapping stop dominating)"); ap.add_argument("--size", default="1080p"); ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
a = ap.parse_args()
w, h, bits = (1920, 1080, 8) if a.size == "1080p" else (3840, 2160, 10)
if a.frames <= 0:
    a.frames = 1536 if a.size == "1080p" else 256
cli = os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")
paths = [os.path.join(a.dir, f"tm_cli_{a.size}_{s}.y4m") for s in ("ref", "dis")]
distinct = 4
pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(distinct)]
for side, p in enumerate(paths):
    with open(p, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 C420{'jpeg' if bits == 8 else 'p10'}\n".encode())
        blobs = [b"FRAME\n" + b"".join(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes() for pl in pr[side]) for pr in pairs]
        for i in range(a.frames):
            f.write(blobs[i % distinct])
print("files:", [round(os.path.getsize(p) / 1e6) for p in paths], "MB", flush=True)
for extra, env in (([], {}), ([], {}), (["--batch", "1", "--no-pipeline"], {}), (["--batch", "4"], {}), (["--batch", "8"], {}), (["--batch", "16"], {}), (["--batch", "32"], {}),
                   ([], {"TM_READER_THREADS": "2"}), ([], {"TM_READER_THREADS": "4"}), ([], {"TM_READER_THREADS": "7"}), ([], {"TM_READER_THREADS": "12"}), ([], {"TM_READER_THREADS": "16"}),
                   (["-m", "psnr", "-m", "msssim"], {})):
    t0 = time.time()
    r = subprocess.run([cli, paths[0], paths[1], "-m", "ssimulacra2", "--output", "json-lines"] + extra, capture_output=True, text=True, env={**os.environ, **env})
    extra = extra + [f"{k}={v}" for k, v in env.items()] or ["(defaults)"]
    dt = time.time() - t0
    line = [l for l in r.stderr.split("\n") if "Processed" in l]
    print(" ".join(extra), "| rc", r.returncode, "| wall %.2fs |" % dt, line[0].strip() if line else r.stderr[-300:], flush=True)
for p in paths:
    os.remove(p)
