#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (may use the oracle).  GPU box: the `turbo-metrics` binary on random planar clips -- random size, bit depth, length, metric set, frame selection -- run in
every host arrangement it has: batches through compute_all with and without the second engine, one pair per launch, the reference's
own loop and the deferred one, two shards on one device, stdin.  Every arrangement must print the same bytes on stdout (JSON lines),
and the first frame's SSIMULACRA2 must be the oracle's.  usage: cli_sweep_soak.py [cases]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
from tm_pkg import tm
from oracle import oracle as O
CLI = os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(5)


def write_y4m(path, frames, w, h, bits):
    cs = "C420jpeg" if bits == 8 else f"C420p{bits}"
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 {cs}\n".encode())
        for planes in frames:
            f.write(b"FRAME\n")
            for pl in planes:
                f.write(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes())


def run(args, env=None, stdin=None):
    r = subprocess.run([CLI] + [str(a) for a in args], capture_output=True, input=stdin, env=None if env is None else dict(os.environ, **env), timeout=300)
    return r.returncode, r.stdout, r.stderr.decode(errors="replace")


t0, bad = time.time(), 0
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    for case in range(cases):
        w, h = int(rng.integers(16, 700)), int(rng.integers(16, 500))
        bits = int(rng.choice([8, 8, 10, 12]))
        n = int(rng.integers(1, 40))
        pairs = [tm.synth.yuv420_pair(w, h, int(rng.integers(0, 1000)), bits) for _ in range(min(n, 6))]
        pr, pd = os.path.join(td, "r.y4m"), os.path.join(td, "d.y4m")
        write_y4m(pr, [pairs[i % len(pairs)][0] for i in range(n)], w, h, bits)
        write_y4m(pd, [pairs[i % len(pairs)][1] for i in range(n)], w, h, bits)
        mets = ["-m", "ssimulacra2"] + (["-m", "psnr"] if rng.random() < 0.5 else []) + (["-m", "ssim"] if w >= 11 and h >= 11 and rng.random() < 0.4 else []) + (
            ["-m", "msssim"] if w >= 176 and h >= 176 and rng.random() < 0.4 else [])
        sel = []
        if rng.random() < 0.3: sel += ["--every", int(rng.integers(2, 5))]
        if rng.random() < 0.3: sel += ["--skip", int(rng.integers(0, max(1, n // 2)))]
        if rng.random() < 0.3: sel += ["--frames", int(rng.integers(1, n + 1))]
        base = [pr, pd] + mets + sel + ["--output", "json-lines"]
        rc0, out0, err0 = run(base)
        arrangements = {"batch1": ["--batch", 1, "--no-pipeline"], "batchN": ["--batch", int(rng.integers(2, 20))], "no_pipeline": ["--no-pipeline"],
                        "loop_reference": ["--loop", "reference"], "loop_deferred": ["--loop", "deferred"], "full_sums": ["--full-sums", "--batch", 3]}
        for name, extra in arrangements.items():
            rc, out, err = run(base + extra)
            if rc != rc0 or out != out0:
                bad += 1
                print(f"MISMATCH case {case} {w}x{h} {bits}-bit {n} frames {mets} {sel}: {name} rc {rc} vs {rc0}\n{err[-300:]}", flush=True)
        rc, out, err = run(base + ["--devices", 2], env={"TM_SHARE_DEVICE": "1"})
        if rc != rc0 or out != out0:
            bad += 1
            print(f"MISMATCH case {case} {w}x{h} {bits}-bit {n} frames {sel}: --devices 2 rc {rc} vs {rc0}\n{err[-300:]}", flush=True)
        # round 6: one process per device + ONE reduce of the score vector (here: ranks sharing the box's GPU, over the launcher's pipes) and
        # 10-bit pictures handed over as 16-bit words instead of packed three to a word
        for name, extra, env in (("ranks", ["--ranks", int(rng.integers(2, 4))], {"TM_SHARE_DEVICE": "1", "TM_RANK_TRANSPORT": "pipe", "TM_RANK_TIMEOUT_S": "300"}),
                                 ("words16", [], {"TM_PACK10": "0"})):
            rc, out, err = run(base + extra, env=env)
            if rc != rc0 or out != out0:
                bad += 1
                print(f"MISMATCH case {case} {w}x{h} {bits}-bit {n} frames {sel}: {name} rc {rc} vs {rc0}\n{err[-300:]}", flush=True)
        rc, out, err = run(["-", pd] + mets + sel + ["--output", "json-lines"], stdin=open(pr, "rb").read())
        if rc != rc0 or out != out0:
            bad += 1
            print(f"MISMATCH case {case} {w}x{h}: reference from stdin rc {rc} vs {rc0}\n{err[-300:]}", flush=True)
        if rc0 == 0 and not sel:  # the first pair against the oracle (fallback matrix by height, like the CLI)
            first = json.loads(out0.decode().splitlines()[0])["ssimulacra2"]
            sr, pit, ch = tm.synth.pack_biplanar(pairs[0][0], w, h, bits); sd, _, _ = tm.synth.pack_biplanar(pairs[0][1], w, h, bits)
            m = 1 if h <= 525 else (2 if h <= 625 else 0)
            lr = O.yuv420_biplanar_to_linear(sr, pit, ch, w, h, 8 if bits == 8 else 16, m); ld = O.yuv420_biplanar_to_linear(sd, pit, ch, w, h, 8 if bits == 8 else 16, m)
            want = O.ssimulacra2_from_linear(lr, ld)[0]
            if abs(first - want) > 1e-9:
                bad += 1
                print(f"ORACLE MISMATCH case {case} {w}x{h} {bits}-bit: {first} vs {want}", flush=True)
print(f"cli sweep: {cases} random clips x 11 host arrangements, mismatches {bad}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
