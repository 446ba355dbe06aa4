#!/usr/bin/env python3
"""bench.py -- SSIMULACRA2 frame-pairs/s on synthetic decoded streams, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p_nv12|4k_p016] [--batch B]

A "step" is one pass of the hot path (ingest -> XYB pyramid -> column pass -> row pass + error maps +
reductions -> 108 sums per pair -> scores) over one batch of B frame pairs whose decoded surfaces are
already resident in HBM.  value = pairs processed by all ranks / wall time of the K timed steps
(barrier + device sync on both sides, max over ranks).  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline      the dominant kernel against the HBM roofline: algorithmic bytes per launch (SURVEY 8d:
                7 f32 per pixel-channel per blur pass = 84 B/px per pass) / its mean launch duration,
                measured with HIP events on the engine's own stream inside the timed region.
  cpu_baseline  the CPU oracle (oracle/tm_oracle.c, a single-thread C restatement of the same
                arithmetic) timed on this host on a bounded sample of the same workload (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, kind, default batch, BASELINE.json config it implements)
    # batch: pairs per step.  The row pass is a few thousand long waves on 1 024 SIMDs, and its tail averages out with more
    # of them: 32 -> 64 slots is worth 2-3 % at 1080p, 16 -> 24 2 % at 4K; memory 15 GB / 23 GB
    "1080p_nv12": (1920, 1080, "nv12", 64, "configs[1]: synthetic 1080p yuv420p frame-pair stream, SSIMULACRA2"),
    "4k_p016": (3840, 2160, "p016", 24, "configs[2]: synthetic 4K yuv420p10 stream, SSIMULACRA2"),
}
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec; 6.29 TB/s measured copy ceiling)


def scale_pixels(w, h):
    tot = 0
    for _ in range(6):
        tot += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=400.0, help="setup, before the W warmup steps: run the path for this long so the device "
                    "reaches its steady clock (the first ~100 ms after idle run 10-15 %% slower); never timed, reported in config")
    ap.add_argument("--workload", default="1080p_nv12", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic pairs generated (cycled over the batch)")
    ap.add_argument("--metrics", default="ssimulacra2", help="comma list: ssimulacra2,psnr,ssim,msssim")
    ap.add_argument("--full-sums", action="store_true", help="compute all 108 per-scale sums like the reference (default: only the 52 with a non-zero weight; same score)")
    ap.add_argument("--no-compare", action="store_true", help="skip the short extra run with the other full_sums setting")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs timed for cpu_baseline (0 = auto, ~15 s)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")

    import torch
    from tm_pkg import tm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # TM_BENCH_BACKEND=gloo (testing only): run the multi-rank path on a box with fewer GPUs than ranks -- the ranks share the
    # devices round-robin and the one collective goes through gloo on host tensors.  The driver's runs use RCCL ("nccl").
    backend = os.environ.get("TM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    cdev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    tm.init_hip(local_rank)

    w, h, kind, default_b, cfg_name = WORKLOADS[args.workload]
    B = args.batch or default_b
    mets = set(args.metrics.split(","))
    metrics = tm.Metrics(ssimulacra2="ssimulacra2" in mets, psnr="psnr" in mets, ssim="ssim" in mets, msssim="msssim" in mets)
    eng = tm.TurboMetrics(w, h, metrics, batch=B)

    # ---- synthetic decoded surfaces, resident in HBM before the timed region (weak scaling: every rank
    # owns its own shard of the stream: pair index = rank*B + slot, cycled over `distinct` generated pairs)
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    distinct = max(1, min(args.distinct, B))
    surfaces = []
    for n in range(distinct):
        (rs, rp, rch), (ds, dp, dch) = gen(w, h, rank * distinct + n)
        surfaces.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
    mk = tm.HwFrame.nv12 if kind == "nv12" else tm.HwFrame.p016
    for slot in range(B):
        (rt, rp, rch), (dt, dp, dch) = surfaces[slot % distinct]
        eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def step():
        eng.compute_async(B)
        eng.sync()

    def timed(steps):
        eng.set_profiling(True)
        eng.stage_ms(reset=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        ms, n = eng.stage_ms(reset=True)
        eng.set_profiling(False)
        return dt, [m / max(n, 1) for m in ms]

    eng.set_full_sums(args.full_sums)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        step()
    for _ in range(args.warmup):
        step()
    elapsed, stage_ms = timed(args.steps)
    modes = eng.job_modes()

    # ---- the single collective of the path: per-frame scores reduced (sum) to rank 0
    scores_local = np.array([eng.scores(i).ssimulacra2 or 0.0 for i in range(B)], np.float64)
    lo, hi = tm.shard.shard_range(world * B, rank, world)  # this rank's block of the stream: [rank*B, (rank+1)*B)
    all_scores = tm.shard.reduce_scores(scores_local, lo, world * B, 1, dist, cdev if dist is not None else "cpu")
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- the same batch with the other setting of full_sums (a few steps, outside the headline timing): by default the
    # engine skips the 56 of 108 per-scale sums whose weight in the reference's table is 0.0; --full-sums computes all
    other = None
    if world == 1 and not args.no_compare and "ssimulacra2" in mets:
        eng.set_full_sums(not args.full_sums)
        step()
        k = max(2, min(5, args.steps))
        dt_o, ms_o = timed(k)
        modes_o = eng.job_modes()
        scores_o = np.array([eng.scores(i).ssimulacra2 or 0.0 for i in range(B)], np.float64)
        other = {"full_sums": not args.full_sums, "value": B * k / dt_o, "ms_per_step": dt_o / k * 1e3,
                 "stage_ms": {"ingest": ms_o[0], "blur_v": ms_o[1], "blur_h": ms_o[2]},
                 "scores_bit_identical": bool(np.array_equal(scores_o, scores_local)), "_modes": modes_o}
        eng.set_full_sums(args.full_sums)

    if rank == 0:
        pairs = world * B * args.steps
        spx = scale_pixels(w, h)
        in_bytes = w * h * 3 // 2 * (1 if kind == "nv12" else 2) * 2  # both frames of a pair
        # ALGORITHMIC bytes per launch (B pairs).  SURVEY 8d model: each blur pass moves 7 f32 per pixel-channel
        # (= 84 B/px summed over the 6 scales) when all five blurred planes of every channel are computed (full_sums).
        # With the zero-weight sums skipped a (scale, channel) image costs 7 (all maps), 4 (edge terms only: mu1, mu2 +
        # ref, dis) or 0 f32 per pixel and pass -- `job_bytes` is what THIS configuration must move.
        # Ingest reads the two surfaces and writes the planar XYB pyramid once (24 B/px for the two sides).
        sizes, ww, hh = [], w, h
        for _ in range(6):
            sizes.append(ww * hh)
            ww, hh = (ww + 1) // 2, (hh + 1) // 2
        units = {0: 0, 1: 4, 2: 7}
        job_bytes = sum(4 * units[int(modes[sc, c])] * sizes[sc] for sc in range(6) for c in range(3))
        model_bytes = 84 * spx

        def roof(ms, nbytes):
            ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes, "achieved_GBs": ach, "frac": ach / HBM_PEAK_GBS}

        # the PMC profiles were taken with SSIMULACRA2 alone: only then do they describe this run
        traffic = load_pmc_traffic(args.workload, B, args.full_sums) if mets == {"ssimulacra2"} else {}
        has_s2 = "ssimulacra2" in mets
        # without SSIMULACRA2 the ingest kernel writes no pyramid (only the u8 planes when SSIM / MS-SSIM ask for them) and the
        # blur kernels are not launched at all
        ingest_bytes = (in_bytes + (24 * spx if has_s2 else 0) + (6 * w * h if mets & {"ssim", "msssim"} else 0)) * B
        per_kernel = {"k_ingest_wave": roof(stage_ms[tm.ffi.TM_STAGE_INGEST], ingest_bytes)}
        if has_s2:
            per_kernel["k_blur_v_jobs"] = roof(stage_ms[tm.ffi.TM_STAGE_BLUR_V], job_bytes * B)
            per_kernel["k_blur_h_jobs_x"] = roof(stage_ms[tm.ffi.TM_STAGE_BLUR_H], job_bytes * B)
        for name in per_kernel:
            per_kernel[name]["traffic"] = traffic.get(name)
        dom = max(("k_blur_v_jobs", "k_blur_h_jobs_x"), key=lambda k: per_kernel[k]["avg_launch_ms"]) if has_s2 else "k_ingest_wave"
        ms_v = per_kernel["k_blur_v_jobs"]["avg_launch_ms"] if has_s2 else 0.0
        ms_h = per_kernel["k_blur_h_jobs_x"]["avg_launch_ms"] if has_s2 else 0.0
        stage_ach = 2 * job_bytes * B / ((ms_v + ms_h) * 1e-3) / 1e9 if ms_v + ms_h > 0 else 0.0
        out = {
            "metric": "ssimulacra2_frame_pairs_per_sec" if has_s2 else "frame_pairs_per_sec",
            "value": pairs / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "baseline_config": cfg_name, "width": w, "height": h, "input": kind,
                       "pairs_per_step_per_gpu": B, "metrics": sorted(mets), "inputs_resident_in_hbm": True, "settle_ms_before_warmup": args.settle_ms,
                       "full_sums": bool(args.full_sums),
                       "parallelism": f"frame-pair sharding x{world}, one RCCL reduce of scores"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": per_kernel[dom]["traffic"],
                         "algorithmic_bytes_per_launch": per_kernel[dom]["algorithmic_bytes_per_launch"],
                         "avg_launch_ms": per_kernel[dom]["avg_launch_ms"],
                         "bytes_model": "SURVEY 8d (84 B/px/pass)" if args.full_sums else
                                        "SURVEY 8d restricted to the planes that carry weight (job table)"},
            "kernels": per_kernel,
            "stages": {"blur_reduce_stage_GBs": stage_ach, "blur_reduce_stage_frac": stage_ach / HBM_PEAK_GBS,
                       "blur_reduce_stage_bytes_per_pair": 2 * job_bytes,
                       "survey_8d_model_bytes_per_pair": 2 * model_bytes,
                       "survey_8d_model_frac": 2 * model_bytes * B / ((ms_v + ms_h) * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_v + ms_h > 0 else 0.0,
                       "full_pipeline_GBs": (pairs / elapsed) * (2 * job_bytes + 24 * spx + in_bytes) / 1e9 / world}
            if has_s2 else {"full_pipeline_GBs": (pairs / elapsed) * (ingest_bytes / B) / 1e9 / world},
            "score_mean": float(np.mean(all_scores)),
        }
        if other is not None:
            mv, mh = other["stage_ms"]["blur_v"], other["stage_ms"]["blur_h"]
            modes_o = other.pop("_modes")
            ob = sum(4 * units[int(modes_o[sc, c])] * sizes[sc] for sc in range(6) for c in range(3))
            other["blur_reduce_stage_bytes_per_pair"] = 2 * ob
            other["blur_reduce_stage_frac"] = 2 * ob * B / ((mv + mh) * 1e-3) / 1e9 / HBM_PEAK_GBS if mv + mh > 0 else 0.0
            out["compare"] = other
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(tm, w, h, kind, args.cpu_pairs)
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def load_pmc_traffic(workload, batch, full_sums=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic_<workload>_b<B>.json, made by
    tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md, which the row pass' known read volume confirms).  Empty when no matching profile exists."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}_b{batch}{'_full' if full_sums else ''}.json")
    if not os.path.exists(path):
        return {}
    d = json.load(open(path))
    out = {}
    for k, v in d.get("kernels", {}).items():
        name = k.replace("void ", "").replace("tmk::", "").split("<")[0]
        f, wr = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if f is not None and wr is not None:
            out[name] = int((2 * f + wr) * 1024)
    return out


def cpu_baseline(tm, w, h, kind, n_pairs):
    """CPU baselines on this host, bounded to 10-15 s in total (reported, never the target):
    the restated reference CPU path (oracle/tm_cpu_path.c == examples/cpu.rs, single-threaded like the original)
    after the same YUV->linear conversion the GPU path applies, timed on 1 thread and with frame-level
    parallelism on the host cores; plus the GPU-arithmetic oracle on 1 thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    bits = 8 if kind == "nv12" else 16
    pairs = [gen(w, h, n) for n in range(2)]

    def one(i, fn):
        (rs, rp, rch), (ds, dp, dch) = pairs[i % len(pairs)]
        lr = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, 0)
        ld = O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, bits, 0)
        return fn(lr, ld)

    if n_pairs <= 0:
        n_pairs = 12 if w * h <= 1920 * 1080 else 3  # ~5 s on one core; with the two other legs the whole baseline is 10-15 s
    t0 = time.perf_counter()
    for i in range(n_pairs):
        one(i, O.cpu_path_score_linear)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(max(1, n_pairs // 2)):
        one(i, lambda a, b: O.ssimulacra2_from_linear(a, b)[0])
    dt_gpu_arith = (time.perf_counter() - t0) / max(1, n_pairs // 2)
    threads = min(os.cpu_count() or 1, 64)
    n_par = threads * 2
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL: real frame-level parallelism
        list(ex.map(lambda i: one(i, O.cpu_path_score_linear), range(n_par)))
    dtp = time.perf_counter() - t0
    return {"value": n_pairs / dt1, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n_pairs} {w}x{h} {kind} pairs, YUV->linear + restated reference CPU path (oracle/tm_cpu_path.c == examples/cpu.rs), 1 thread",
            "seconds": dt1, "host_cpus": os.cpu_count(),
            "all_cores": {"value": n_par / dtp, "cores": threads, "sample": f"{n_par} pairs, one pair per worker thread", "seconds": dtp},
            "gpu_arithmetic_oracle_1thread_pairs_per_s": 1.0 / dt_gpu_arith}


if __name__ == "__main__":
    main()
