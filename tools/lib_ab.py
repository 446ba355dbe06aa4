#!/usr/bin/env python3
"""GPU box: A/B of two laboratory builds of the engine library (the committed one, lab/libturbometrics_hip_lab.so, against a build from `make -C turbo-metrics_amd/csrc ab
EXPFLAGS=... EXPNAME=...`).  Each arm runs in a process of its own (TM_HIP_LIB), the arms alternate A B A B ..., every run = one
engine, bench.py's pairs per step (bench.WORKLOADS: 128 1080p NV12 / 48 4K P016; --batch B overrides), 32 distinct pairs, 400 ms of
settling, 60 timed steps with stage events.
Reports per arm: step ms, pairs/s, stage ms [ingest, col, row, edge, finish] and the scores' checksum (the arms must agree).

    python tools/lib_ab.py A=turbo-metrics_amd/lab/libturbometrics_hip_lab.so B=build_exp/libtm_x.so [--rounds 3] [--workload 4k_p016] [--batch B]
An arm may name an engine variant of the same library instead of another build: B=turbo-metrics_amd/lab/libturbometrics_hip_lab.so@0x2000
(tm_engine_set_variant: here TM_VARIANT_UPPER_KERNEL).
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from tm_pkg import tm
wl = %(wl)r
w, h, gen, mk = (1920, 1080, tm.synth.nv12_pair, tm.HwFrame.nv12) if wl == "1080p_nv12" else (3840, 2160, tm.synth.p016_pair, tm.HwFrame.p016)
B = %(batch)d
tm.init_hip(0)
ND = 32 if wl == "1080p_nv12" else 2
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(8) as ex:
    host = list(ex.map(lambda n: gen(w, h, n), range(ND)))
pairs = [((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)) for (rs, rp, rch), (ds, dp, dch) in host]
eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
eng.set_variant(%(variant)d)
for slot in range(B):
    (rt, rp, rch), (dt, dp, dch) = pairs[slot %% ND]
    eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    eng.compute_async(); eng.sync()
eng.set_profiling(True); eng.stage_ms(reset=True)
K = 60
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K):
    eng.compute_async(); eng.sync()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ms, n = eng.stage_ms(reset=True)
sc = np.array([s.ssimulacra2 for s in eng.scores_batch(B)])
print(json.dumps({"step_ms": dt / K * 1e3, "pairs_per_s": B * K / dt, "stage_ms": [m / n for m in ms], "sha": hashlib.sha256(sc.tobytes()).hexdigest()[:12]}))
'''


def main():
    arms = [a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "1080p_nv12"
    sys.path.insert(0, ROOT)
    import bench  # the pairs per step of the committed bench and PMC figures
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else bench.WORKLOADS[wl][3]
    res = {name: [] for name, _ in arms}
    for r in range(rounds):
        for name, lib in arms:
            lib, _, variant = lib.partition("@")
            env = dict(os.environ, TM_HIP_LIB=os.path.join(ROOT, lib))
            p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "wl": wl, "batch": batch, "variant": int(variant or "0", 0)}], capture_output=True, text=True, env=env, timeout=600)
            if p.returncode != 0:
                print(name, "FAILED", p.stderr[-800:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[name].append(d)
            print(f"{name} round {r}: step {d['step_ms']:.3f} ms  {d['pairs_per_s']:.0f} pairs/s  stages " + " ".join(f"{m:.3f}" for m in d["stage_ms"]) + f"  sha {d['sha']}", flush=True)
    print("# medians")
    for name, _ in arms:
        v = sorted(res[name], key=lambda d: d["step_ms"])
        if v:
            m = v[len(v) // 2]
            print(f"{name}: step {m['step_ms']:.3f} ms  {m['pairs_per_s']:.0f} pairs/s  ingest {m['stage_ms'][0]:.3f} ms  sha {m['sha']}")
    shas = {d["sha"] for v in res.values() for d in v}
    print("# scores identical across arms:", len(shas) == 1)


if __name__ == "__main__":
    main()
