#!/usr/bin/env python3
"""bench.py -- SSIMULACRA2 frame-pairs/s on synthetic decoded streams, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p_nv12|4k_p016] [--batch B] [--metrics ...]

A "step" is one pass of the hot path (ingest -> XYB pyramid -> column pass -> row pass + error maps +
reductions -> 108 sums per pair -> scores) over one batch of B frame pairs whose decoded surfaces are
already resident in HBM.  value = pairs processed by all ranks / wall time of the K timed steps
(barrier + device sync on both sides, max over ranks).  Prints ONE JSON line on rank 0.

--gpus N > 1 without a torchrun environment: this process -- before it imports torch or touches the GPU --
starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, turbo-metrics_amd/launch.py),
forwards rank 0's line and exits non-zero when any rank fails.  Under torchrun (WORLD_SIZE set) it is a rank.

Objects on the line (besides the contract fields, which describe the headline workload = BASELINE configs[1]):
  roofline      the dominant kernel against the HBM roofline: algorithmic bytes per launch / its mean launch
                duration, measured with HIP events on the engine's own stream inside the timed region.
  workloads     the other configurations BASELINE.json's metric names, each with value / ms_per_step / roofline:
                4K P016 (configs[2]) and the fused PSNR + MS-SSIM + SSIMULACRA2 pass at 1080p and 4K (configs[4]);
                shorter runs of the same code (min(K, 10) steps).  Skipped with --no-extras or an explicit --workload.
  fixed_stream  BASELINE configs[3] / SURVEY 8d config 4: a stream of 2048 1080p pairs (fixed total = strong scaling),
                contiguous shards, wall clock from the first submit to rank 0 holding all 2048 scores (one reduce);
                `long` inside it: the same with 16 384 pairs, so that every rank still has >= 0.15 s of kernels at 8 GPUs.
  host_fed      SURVEY 8d config 2 (ii): the same pairs uploaded from page-locked host memory for every step
                (two engines ping-pong: the upload of batch k+1 overlaps the kernels of batch k).  Never `value`.
  batch_curve   the headline workload at 1, 2, 4, 8, 16, 32, 64, 128 pairs per launch (the reference's compute_one is ONE pair per
                call): pairs/s, ms per step, engine memory; for 1 .. 8 pairs per launch also with 2 launches IN FLIGHT (two
                engines taking turns, each waited for only before its next launch: what an asynchronous caller of the C ABI
                -- or compute_all -- gets at the same call granularity).  N = 1 only.
  cli_end_to_end  the C++ `turbo-metrics` binary on Y4M clips in tmpfs (1080p 8-bit, 4K 10-bit): file -> pinned ring -> upload
                -> SSIMULACRA2 -> JSON lines, its own "Processed ... fps" figure.  N = 1 only.  Never `value`.
  cpu_baseline  the restated reference CPU path timed on this host on a bounded sample (rank 0, N=1).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, kind, default batch, BASELINE.json config it implements)
    # batch: pairs per step.  The row pass is a few thousand long waves on 1 024 SIMDs, and its tail averages out with more
    # of them: 32 -> 64 slots is worth 2-3 % at 1080p, 64 -> 128 another 2.5 % (192: +1 %, 256: nothing more); 4K: 24 -> 36 -> 48
    # slots +3.0 / +3.7 %, 72 nothing more (profiles/r05t_batch_size.log).  Memory 31 GB / 46 GB of the 288 GB.
    "1080p_nv12": (1920, 1080, "nv12", 128, "configs[1]: synthetic 1080p yuv420p frame-pair stream, SSIMULACRA2"),
    "4k_p016": (3840, 2160, "p016", 48, "configs[2]: synthetic 4K yuv420p10 stream, SSIMULACRA2"),
}
SHARED_DEVICE_BATCH = {"1080p_nv12": 64, "4k_p016": 24}  # ranks that share ONE device (the gloo rehearsal on a 1-GPU box): half the slots each
FUSED = "psnr,msssim,ssimulacra2"  # BASELINE configs[4]
EXTRAS = [("4k_p016", "ssimulacra2"), ("1080p_nv12", FUSED), ("4k_p016", FUSED)]
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec; 6.29 TB/s measured copy ceiling)
STREAM_PAIRS = 2048
STREAM_PAIRS_LONG = 16384  # the strong leg's longer companion: 2048 pairs per rank at 8 GPUs (~0.17 s of kernels)
BATCH_CURVE = (1, 2, 4, 8, 16, 32, 64, 128)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=float, default=400.0, help="setup, before the W warmup steps: run the path for this long so the device "
                    "reaches its steady clock (the first ~100 ms after idle run 10-15 %% slower); never timed, reported in config")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="run only this workload (default: 1080p_nv12 headline + the extras)")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic pairs generated (cycled over the batch; SURVEY 8d config 2: 32)")
    ap.add_argument("--metrics", default="ssimulacra2", help="comma list: ssimulacra2,psnr,ssim,msssim")
    ap.add_argument("--full-sums", action="store_true", help="compute all 108 per-scale sums like the reference (default: only the 52 with a non-zero weight; same score)")
    ap.add_argument("--no-compare", action="store_true", help="skip the short extra run with the other full_sums setting")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only (no `workloads`, `fixed_stream`, `host_fed` objects)")
    ap.add_argument("--stream-pairs", type=int, default=None, help=f"pairs of the fixed-size stream (strong scaling leg; default {STREAM_PAIRS} "
                    f"plus a second leg of {STREAM_PAIRS_LONG}); given explicitly the leg also runs under --no-extras")
    ap.add_argument("--no-cli", action="store_true", help="skip the cli_end_to_end leg")
    ap.add_argument("--edge-beside", type=int, default=1, choices=(0, 1, 2), help="measurement: 0 = the fused kernel of the edge-only jobs behind the row pass "
                    "(every kernel alone on the chip: per-kernel profiles), 1 = beside the two passes (the default, what ships)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail-file", default=None, help="where the full record goes (default: bench_detail.json beside bench.py, and a copy "
                    "under gpurun_out/ when that directory exists); the stdout line is the compact one (<= 4 KB)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs timed for cpu_baseline (0 = auto, ~15 s)")
    return ap.parse_args()


def scale_sizes(w, h):
    out = []
    for _ in range(6):
        out.append(w * h)
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def csrc_sha16():
    """content hash of the kernel sources: a PMC traffic profile describes exactly one version of them"""
    d = os.path.join(ROOT, "turbo-metrics_amd", "csrc")
    hh = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".inc")):
            hh.update(name.encode())
            hh.update(open(os.path.join(d, name), "rb").read())
    return hh.hexdigest()[:16]


def launch_module():
    """turbo-metrics_amd/launch.py loaded on its own: no numpy, no torch, nothing that touches the GPU"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tm_launch_standalone", os.path.join(ROOT, "turbo-metrics_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Ctx:
    """what every leg needs: rank layout, torch, the package, the (optional) process group"""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"WORLD_SIZE={self.world} but --gpus {args.gpus}")
        # this rank's share of the CPUs the job may use (cgroup quota / affinity), set before numpy and torch load their thread pools
        # (launch.py on its own: importing the package pulls in numpy)
        self.cpu_share = launch_module().cap_rank_threads(self.world)
        from tm_pkg import tm
        import numpy as np
        import torch
        torch.set_num_threads(self.cpu_share)
        self.np, self.torch, self.tm = np, torch, tm
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product has no CPU path")
        # TM_BENCH_BACKEND=gloo (testing only): run the multi-rank path on a box with fewer GPUs than ranks -- the ranks share the
        # devices round-robin and the one collective goes through gloo on host tensors.  The driver's runs use RCCL ("nccl").
        self.backend = os.environ.get("TM_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if self.backend == "nccl" and ndev < self.world:
            raise SystemExit(f"--gpus {self.world} but only {ndev} device(s) visible")
        self.shared_device = self.backend != "nccl" and self.world > max(1, ndev)
        if self.backend != "nccl":
            self.local_rank %= max(1, ndev)
        self.cdev = "cuda" if self.backend == "nccl" else "cpu"
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        # TM_BENCH_FORCE_DIST=1 (testing only): a process group of ONE rank under torchrun-style variables -- every collective of
        # the path (barrier, all_reduce(MAX), reduce(SUM)) then really goes through RCCL on a 1-GPU box
        if self.world > 1 or os.environ.get("TM_BENCH_FORCE_DIST") == "1":
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
            self.dist = dist
        tm.init_hip(self.local_rank)
        # the rank's threads (surface generators, page-locked allocations) next to its device, like the CLI's
        self.numa_node = tm.ffi.lib().tm_device_numa_node(self.local_rank)
        if self.numa_node < 0:  # (torch's bundled HIP runtime answers -1: ask sysfs for the PCI device's node)
            try:
                pr = torch.cuda.get_device_properties(self.local_rank)
                self.numa_node = tm.launch.pci_numa_node(int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
            except Exception:  # noqa: BLE001
                pass
        self.cpus_bound = tm.launch.bind_to_numa_node(self.numa_node)
        self._surfaces = {}

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def device_mem_used_gb(self):
        import ctypes as C
        free, total = C.c_size_t(), C.c_size_t()
        if self.tm.ffi.lib().tm_device_mem_info(C.byref(free), C.byref(total)) != 0:
            return None
        return round((total.value - free.value) / 1e9, 2)

    def min_max_over_ranks(self, x):
        """[min, max] of a per-rank figure: a straggler shows on the one line the driver keeps"""
        return self.tm.shard.min_max_over_ranks(x, self.dist, self.cdev)

    def surfaces(self, name, distinct, pinned=False):
        """`distinct` synthetic pairs of the workload (the same on every rank: pair i of a stream has content i % distinct),
        resident in HBM -- or in page-locked host memory for the host-fed leg.  Cached: 4K pairs take seconds to generate."""
        key = (name, distinct, pinned)
        if key not in self._surfaces:
            w, h, kind, _, _ = WORKLOADS[name]
            gen = self.tm.synth.nv12_pair if kind == "nv12" else self.tm.synth.p016_pair
            host = self._surfaces.get((name, distinct, "host"))
            if host is None:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max(1, min(8, distinct, self.cpu_share))) as ex:  # numpy releases the GIL in the heavy parts
                    host = self._surfaces[(name, distinct, "host")] = list(ex.map(lambda n: gen(w, h, n), range(distinct)))
            if pinned:
                # a decoder's surface pool: the page-locked surfaces of a side lie back to back -- what the engine copies of one (coded_height
                # luma rows + ceil(h / 2) CbCr rows) ends where the next begins, so two frames of 1080p can share a DMA (tm_engine.hip, queue_copy)
                (rs0, rp0, rch0) = host[0][0]
                frame = rp0 * (rch0 + (h + 1) // 2)
                pools = [self.torch.empty(frame * distinct + len(rs0), dtype=self.torch.uint8).pin_memory() for _ in range(2)]
                out = []
                for k, ((rs, rp, rch), (ds, dp, dch)) in enumerate(host):
                    views = []
                    for side, surf in enumerate((rs, ds)):
                        v = pools[side][k * frame:k * frame + len(surf)]
                        v[:frame].copy_(self.torch.from_numpy(surf[:frame]))  # (what lies behind the CbCr rows of a surface is never read)
                        views.append(v)
                    out.append(((views[0], rp, rch), (views[1], dp, dch)))
                self._surfaces[key] = out
                self._pools = getattr(self, "_pools", []) + pools
            else:
                put = lambda a: self.torch.from_numpy(a).cuda()
                self._surfaces[key] = [((put(rs), rp, rch), (put(ds), dp, dch)) for (rs, rp, rch), (ds, dp, dch) in host]
            self.torch.cuda.synchronize()
        return self._surfaces[key]

    def fill_slots(self, eng, name, distinct, first_pair, n, pinned=False):
        """slots [0, n) <- pairs first_pair .. first_pair + n - 1 of the stream"""
        kind = WORKLOADS[name][2]
        mk = self.tm.HwFrame.nv12 if kind == "nv12" else self.tm.HwFrame.p016
        sf = self.surfaces(name, distinct, pinned)
        for slot in range(n):
            (rt, rp, rch), (dt, dp, dch) = sf[(first_pair + slot) % distinct]
            eng.set_pair(slot, mk(rt, rp, rch), mk(dt, dp, dch))


def load_pmc_traffic(workload, batch, full_sums=False):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic_<workload>_b<B>.json, made by
    tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md, which the row pass' known read volume confirms).  A profile describes ONE version of the kernels: it
    carries the content hash of csrc/ it was taken with, and a profile of another version is not reported."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}_b{batch}{'_full' if full_sums else ''}.json")
    if not os.path.exists(path):
        return {}, "no PMC profile for this workload / batch"
    d = json.load(open(path))
    if d.get("csrc_sha16") != csrc_sha16():
        return {}, f"PMC profile is of another kernel version (csrc {d.get('csrc_sha16')}, this run {csrc_sha16()})"
    out = {}
    for k, v in d.get("kernels", {}).items():
        name = k.replace("void ", "").replace("tmk::", "").split("<")[0]
        f, wr = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
        if f is not None and wr is not None:
            out[name] = int((2 * f + wr) * 1024)
    return out, None


def run_workload(ctx, args, name, mets, B, steps, warmup, settle_ms, compare, keep_engine=False):
    """K timed steps of one workload on every rank; returns (result dict on every rank, engine or None)."""
    tm, np, torch = ctx.tm, ctx.np, ctx.torch
    w, h, kind, _, cfg_name = WORKLOADS[name]
    metrics = tm.Metrics(ssimulacra2="ssimulacra2" in mets, psnr="psnr" in mets, ssim="ssim" in mets, msssim="msssim" in mets)
    eng = tm.TurboMetrics(w, h, metrics, batch=B)
    distinct = max(1, min(args.distinct if w * h <= 1920 * 1080 else 2, B))
    # weak scaling: every rank owns its own block of the stream, [rank * B, (rank + 1) * B)
    ctx.fill_slots(eng, name, distinct, ctx.rank * B, B)
    torch.cuda.synchronize()

    def step():
        eng.compute_async(B)
        eng.sync()

    def timed(k):
        eng.set_profiling(True)
        eng.stage_ms(reset=True)
        ctx.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
        ctx.barrier()
        dt = time.perf_counter() - t0
        ms, n = eng.stage_ms(reset=True)
        eng.set_profiling(False)
        timed.own_s = own
        return dt, [m / max(n, 1) for m in ms]

    eng.set_full_sums(args.full_sums)
    if args.edge_beside != 1:
        eng.debug_set_edge_beside(args.edge_beside)
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < settle_ms:
        step()
    for _ in range(warmup):
        step()
    elapsed, stage_ms = timed(steps)
    elapsed = ctx.max_over_ranks(elapsed)
    per_rank = ctx.min_max_over_ranks(B * steps / timed.own_s)
    modes = eng.job_modes()
    has_s2 = "ssimulacra2" in mets
    sc = eng.scores_batch(B)
    scores_local = np.array([(s.ssimulacra2 if has_s2 else s.psnr) or 0.0 for s in sc], np.float64)
    # the single collective of the path: per-frame scores reduced (sum into zeros) to rank 0
    rdev = ctx.cdev if ctx.dist is not None else "cpu"
    reduce_ms = []
    for _ in range(2 if ctx.dist is not None else 1):  # the first reduce of a process group also builds its channels: timed apart
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        all_scores = tm.shard.reduce_scores(scores_local, ctx.rank * B, ctx.world * B, 1, ctx.dist, rdev)
        torch.cuda.synchronize()
        reduce_ms.append(ctx.max_over_ranks(time.perf_counter() - t_r) * 1e3)

    alone = None
    if has_s2 and eng.uses_fused_edge(B) and args.edge_beside == 1:
        # the fused kernel of the edge-only jobs runs BESIDE the two blur passes (second stream): the stage times above overlap.
        # A few steps with it BEHIND the row pass instead (outside the headline timing) time every kernel alone on the chip.
        eng.debug_set_edge_beside(0)
        step()
        k = max(2, min(5, steps))
        dt_a, ms_a = timed(k)
        eng.debug_set_edge_beside(1)
        step()
        alone = {"ms_per_step": dt_a / k * 1e3, "value": B * k / dt_a, "stage_ms": ms_a}

    other = None
    if compare and ctx.world == 1 and has_s2:
        # the same batch with the other setting of full_sums (a few steps, outside the headline timing): by default the engine
        # skips the 56 of 108 per-scale sums whose weight in the reference's table is 0.0; --full-sums computes all
        eng.set_full_sums(not args.full_sums)
        step()
        k = max(2, min(5, steps))
        dt_o, ms_o = timed(k)
        modes_o = eng.job_modes()
        scores_o = np.array([s.ssimulacra2 or 0.0 for s in eng.scores_batch(B)], np.float64)
        other = {"full_sums": not args.full_sums, "value": B * k / dt_o, "ms_per_step": dt_o / k * 1e3,
                 "stage_ms": {"ingest": ms_o[0], "blur_v": ms_o[1], "blur_h": ms_o[2], "edge_fused": ms_o[tm.ffi.TM_STAGE_EDGE]},
                 "scores_bit_identical": bool(np.array_equal(scores_o, scores_local)), "_modes": modes_o, "_fused": bool(eng.uses_fused_edge(B))}
        eng.set_full_sums(args.full_sums)

    pairs = ctx.world * B * steps
    sizes = scale_sizes(w, h)
    spx = sum(sizes)
    in_bytes = w * h * 3 // 2 * (1 if kind == "nv12" else 2) * 2  # both frames of a pair
    # ALGORITHMIC bytes per launch (B pairs).  SURVEY 8d model: each blur pass moves 7 f32 per pixel-channel (= 84 B/px summed
    # over the 6 scales) when all five blurred planes of every channel are computed (full_sums).  With the zero-weight sums
    # skipped a (scale, channel) image costs 7 (all maps), 4 (edge terms only: mu1, mu2 + ref, dis) or 0 f32 per pixel and
    # pass -- `job_bytes` is what THIS configuration must move.  Ingest reads the two surfaces and writes the planar XYB
    # pyramid once (24 B/px for the two sides).  The SSIM / MS-SSIM stage reads the u8-quantised planes (6 B/px per pair) and,
    # for MS-SSIM, builds and reads back the dyadic pyramid of box sums (u16, scales 1-4: 2 x 0.332 x 2 B per sample).
    # Launches of >= 400 bands of EDGE planes (6 pairs of 1080p, 3 of 4K) run the edge-only jobs in ONE kernel (k_blur_edge_fused) that reads the {ref, dis} plane
    # once (2 f32 per pixel) and writes nothing: the two passes then move the bytes of the FULL jobs only.
    units = {0: 0, 1: 4, 2: 7}
    fused_edge = bool(has_s2 and eng.uses_fused_edge(B))
    pass_units = {0: 0, 1: 0 if fused_edge else 4, 2: 7}
    job_bytes = sum(4 * pass_units[int(modes[s, c])] * sizes[s] for s in range(6) for c in range(3))
    edge_bytes = sum(4 * 2 * sizes[s] for s in range(6) for c in range(3) if int(modes[s, c]) == 1) if fused_edge else 0
    model_bytes = 84 * spx
    has_ssim = bool(mets & {"ssim", "msssim"})
    ssim_bytes = int((6 + (6 * 0.332 * 2 * 2 if "msssim" in mets else 0)) * w * h)

    def roof(ms, nbytes):
        ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": nbytes, "achieved_GBs": ach, "frac": ach / HBM_PEAK_GBS}

    # the PMC profiles were taken with SSIMULACRA2 alone: only then do they describe this run
    traffic, traffic_note = load_pmc_traffic(name, B, args.full_sums) if mets == {"ssimulacra2"} else ({}, "PMC profiles exist for SSIMULACRA2 alone")
    F = tm.ffi
    # without SSIMULACRA2 the ingest kernel writes no pyramid (only the u8 planes when SSIM / MS-SSIM ask for them) and the
    # blur kernels are not launched at all
    ingest_bytes = (in_bytes + (24 * spx if has_s2 else 0) + (6 * w * h if has_ssim else 0)) * B
    ingest_name = "k_ingest_rows" if kind in ("nv12", "p016") else "k_ingest_wave"  # (since round 6 the 4:2:0 kernel finishes all six levels itself)
    per_kernel = {ingest_name: roof(stage_ms[F.TM_STAGE_INGEST], ingest_bytes)}
    if has_s2:
        per_kernel["k_blur_v_jobs"] = roof(stage_ms[F.TM_STAGE_BLUR_V], job_bytes * B)
        per_kernel["k_blur_h_jobs_x"] = roof(stage_ms[F.TM_STAGE_BLUR_H], job_bytes * B)
        if fused_edge:
            per_kernel["k_blur_edge_fused"] = roof(stage_ms[F.TM_STAGE_EDGE], edge_bytes * B)
            per_kernel["k_blur_edge_fused"]["bound"] = "instruction issue when alone on the chip (both recurrences, the edge maps and their f64 sums of an edge-only job in one kernel: ~1 550 instructions per 32 x 32 pixel pairs against 8 KB of input); algorithmic bytes = the input read once -- its PMC traffic is twice that (10 of every 42 input rows read again by the next band, 6 B per pixel pair of state hand-off words)"
    if has_ssim:
        per_kernel["k_ssim_stage"] = roof(stage_ms[F.TM_STAGE_SSIM], ssim_bytes * B)
        per_kernel["k_ssim_stage"]["bound"] = "valu (11x11 separable window of 4 quantities: 88 fused multiply-adds per window and channel; HBM fraction is informative only)"
    for kn in per_kernel:
        per_kernel[kn]["traffic"] = traffic.get(kn)
    if alone is not None:  # the same kernels with nothing beside them
        ma = alone.pop("stage_ms")
        alone["kernels"] = {"k_blur_v_jobs": roof(ma[F.TM_STAGE_BLUR_V], job_bytes * B), "k_blur_h_jobs_x": roof(ma[F.TM_STAGE_BLUR_H], job_bytes * B),
                            "k_blur_edge_fused": roof(ma[F.TM_STAGE_EDGE], edge_bytes * B)}
        alone["note"] = "k_blur_edge_fused behind the row pass on the engine's stream (tm_engine_debug_set_edge_beside 0): every kernel alone on the chip"
    dom = max(("k_blur_v_jobs", "k_blur_h_jobs_x"), key=lambda k: per_kernel[k]["avg_launch_ms"]) if has_s2 else ingest_name
    ms_v = per_kernel["k_blur_v_jobs"]["avg_launch_ms"] if has_s2 else 0.0
    ms_h = per_kernel["k_blur_h_jobs_x"]["avg_launch_ms"] if has_s2 else 0.0
    # the blur + reduce stage = the two passes and the fused kernel BESIDE them: the bytes they have to move over the span from the
    # end of the ingest stage to the end of the row pass (+ what the finisher then still waits for the fused kernel: that wait is
    # in the SSIM stage's events; without SSIM metrics that stage is the wait and a 10-us finisher)
    ms_e = (stage_ms[F.TM_STAGE_SSIM] if not has_ssim else 0.0) if fused_edge else 0.0
    stage_bytes = 2 * job_bytes + edge_bytes
    stage_ach = stage_bytes * B / ((ms_v + ms_h + ms_e) * 1e-3) / 1e9 if ms_v + ms_h > 0 else 0.0
    bytes_model = ("SURVEY 8d (84 B/px/pass)" if args.full_sums else
                   "SURVEY 8d restricted to the planes that carry weight (job table)" + ("; the edge-only jobs run in k_blur_edge_fused (algorithmic bytes: their input, read once)" if fused_edge else ""))
    if alone is not None:
        # The launches that dominate the step run CONCURRENTLY: the fused kernel of the edge-only jobs is a persistent launch beside the
        # column pass and the row pass.  One kernel's bytes over its own duration would leave out what the others move through HBM
        # meanwhile (column pass: its 7.4 GB over 1.9 ms shared = 0.49, over 1.45 ms alone = 0.64) -- the roofline entry is the GROUP:
        # the three kernels' algorithmic bytes over the span from the end of the ingest stage to the end of the row pass, measured
        # with the HIP events of the timed steps; every member's own figures, beside the others and alone, are listed.
        grp = ("k_blur_v_jobs", "k_blur_h_jobs_x", "k_blur_edge_fused")
        gtraffic = sum(traffic[k] for k in grp) if all(traffic.get(k) for k in grp) else None
        roofline = {"bound": "hbm", "kind": "group", "kernel": "k_blur_v_jobs + k_blur_h_jobs_x + k_blur_edge_fused (one concurrent group)",
                    "achieved": stage_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": stage_ach / HBM_PEAK_GBS,
                    "traffic": gtraffic, "traffic_note": traffic_note,
                    "algorithmic_bytes_per_launch": stage_bytes * B, "avg_launch_ms": ms_v + ms_h + ms_e, "bytes_model": bytes_model,
                    "members": {k: {"algorithmic_bytes_per_launch": per_kernel[k]["algorithmic_bytes_per_launch"], "traffic": per_kernel[k]["traffic"],
                                    "avg_launch_ms": per_kernel[k]["avg_launch_ms"], "frac": per_kernel[k]["frac"],
                                    "avg_launch_ms_alone": alone["kernels"][k]["avg_launch_ms"], "frac_alone": alone["kernels"][k]["frac"]} for k in grp},
                    "longest_hbm_bound_member": dom, "frac_alone": alone["kernels"][dom]["frac"], "avg_launch_ms_alone": alone["kernels"][dom]["avg_launch_ms"],
                    "region_frac": stage_ach / HBM_PEAK_GBS, "region_ms": ms_v + ms_h + ms_e}
    else:
        roofline = {"bound": "hbm", "kind": "kernel", "kernel": dom, "achieved": per_kernel[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": per_kernel[dom]["traffic"], "traffic_note": traffic_note,
                    "algorithmic_bytes_per_launch": per_kernel[dom]["algorithmic_bytes_per_launch"],
                    "avg_launch_ms": per_kernel[dom]["avg_launch_ms"], "bytes_model": bytes_model}
    res = {
        "value": pairs / elapsed,
        "unit": "frame-pairs/s",
        "ms_per_step": elapsed / steps * 1e3,
        "steps": steps,
        "config": {"workload": name, "baseline_config": cfg_name if mets == {"ssimulacra2"} else "configs[4] on this many GPUs: PSNR + MSSSIM + SSIMULACRA2 fused pass" if mets == set(FUSED.split(",")) else cfg_name + " + " + ",".join(sorted(mets)),
                   "width": w, "height": h, "input": kind, "pairs_per_step_per_gpu": B, "metrics": sorted(mets),
                   "inputs_resident_in_hbm": True, "distinct_pairs_cycled": distinct, "settle_ms_before_warmup": settle_ms, "full_sums": bool(args.full_sums),
                   "engine_mem_GB": round(eng.mem_usage() / 1e9, 2),
                   "parallelism": f"frame-pair sharding x{ctx.world}, one {'RCCL' if ctx.backend == 'nccl' else ctx.backend} reduce of scores"},
        "roofline": roofline,
        "kernels": per_kernel,
        **({"kernels_alone": alone} if alone is not None else {}),
        "stages": {"blur_reduce_stage_GBs": stage_ach, "blur_reduce_stage_frac": stage_ach / HBM_PEAK_GBS,
                   "blur_reduce_stage_bytes_per_pair": stage_bytes,
                   "blur_reduce_stage_ms": ms_v + ms_h + ms_e, "edge_jobs_fused": fused_edge,
                   # what the stage really moves through HBM (PMC traffic of its kernels: the fused kernel re-reads 10 of every 42 input
                   # rows and hands the column recurrence's state from band to band through memory) over the same span
                   **({"blur_reduce_stage_traffic_bytes_per_pair": sum(traffic[k] for k in ("k_blur_v_jobs", "k_blur_h_jobs_x") + (("k_blur_edge_fused",) if fused_edge else ())) // B,
                       "blur_reduce_stage_traffic_frac": sum(traffic[k] for k in ("k_blur_v_jobs", "k_blur_h_jobs_x") + (("k_blur_edge_fused",) if fused_edge else ())) / ((ms_v + ms_h + ms_e) * 1e-3) / 1e9 / HBM_PEAK_GBS}
                      if all(traffic.get(k) for k in ("k_blur_v_jobs", "k_blur_h_jobs_x") + (("k_blur_edge_fused",) if fused_edge else ())) and ms_v + ms_h > 0 else {}),
                   "survey_8d_model_bytes_per_pair": 2 * model_bytes,
                   "survey_8d_model_frac": 2 * model_bytes * B / ((ms_v + ms_h + ms_e) * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_v + ms_h > 0 else 0.0,
                   "full_pipeline_GBs": (pairs / elapsed) * (stage_bytes + 24 * spx + in_bytes) / 1e9 / ctx.world}
        if has_s2 else {"full_pipeline_GBs": (pairs / elapsed) * (ingest_bytes / B) / 1e9 / ctx.world},
        "score_mean": float(np.mean(all_scores)) if all_scores is not None else None,
        # per-rank pairs/s over the rank's own K steps (before the closing barrier), [slowest, fastest]; the one collective of the
        # path, max over ranks: steady state, and the first call (which builds the process group's channels)
        "per_rank": per_rank,
        "stage_ms": [stage_ms[F.TM_STAGE_INGEST], stage_ms[F.TM_STAGE_BLUR_V], stage_ms[F.TM_STAGE_BLUR_H], stage_ms[F.TM_STAGE_EDGE], stage_ms[F.TM_STAGE_SSIM]],
        "reduce_ms": reduce_ms[-1], "reduce_first_ms": reduce_ms[0],
        "rank_cpu": {"threads": ctx.cpu_share, "numa_node": ctx.numa_node, "cpus_bound": ctx.cpus_bound},
        "device_mem_used_GB": ctx.device_mem_used_gb(),  # of rank 0's device, every process on it included, while the engine is alive
    }
    if other is not None:
        mv, mh, me = other["stage_ms"]["blur_v"], other["stage_ms"]["blur_h"], other["stage_ms"]["edge_fused"]
        modes_o = other.pop("_modes")
        fo = other.pop("_fused")
        ob = sum(4 * (2 if fo and int(modes_o[s, c]) == 1 else 2 * units[int(modes_o[s, c])]) * sizes[s] for s in range(6) for c in range(3))
        other["blur_reduce_stage_bytes_per_pair"] = ob
        other["blur_reduce_stage_frac"] = ob * B / ((mv + mh + me) * 1e-3) / 1e9 / HBM_PEAK_GBS if mv + mh > 0 else 0.0
        res["compare"] = other
    if keep_engine:
        return res, eng, distinct
    eng.close()
    return res, None, distinct


def run_fixed_stream(ctx, eng, name, B, distinct, total):
    """SURVEY 8d config 4 / BASELINE configs[3]: a stream of `total` pairs (pair i has content i % distinct), contiguous
    shards (tm.shard.shard_range), each rank runs its shard in batches of B on its own GPU; ONE reduce(sum) of the zero-padded
    score vector; wall clock from the first submit until rank 0 holds every score.  Strong scaling: total is fixed."""
    tm, np, torch = ctx.tm, ctx.np, ctx.torch
    lo, hi = tm.shard.shard_range(total, ctx.rank, ctx.world)
    first_mod = None
    if ctx.dist is not None:  # setup: the first reduce of a process group builds its channels (tens of ms over RCCL) -- not the stream's work
        tm.shard.reduce_scores(np.zeros(hi - lo, np.float64), lo, total, 1, ctx.dist, ctx.cdev)
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local = []
    for b0 in range(lo, hi, B):
        n = min(B, hi - b0)
        if b0 % distinct != first_mod:  # descriptors only change when the batch starts at another phase of the cycle
            ctx.fill_slots(eng, name, distinct, b0, B)
            first_mod = b0 % distinct
        eng.compute_async(n)
        eng.sync()
        local.extend(s.ssimulacra2 for s in eng.scores_batch(n))
    all_scores = tm.shard.reduce_scores(np.array(local, np.float64), lo, total, 1, ctx.dist, ctx.cdev if ctx.dist is not None else "cpu")
    torch.cuda.synchronize()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    if ctx.rank != 0:
        return None
    sc = all_scores.ravel()
    # every rank computed a disjoint block and the reduce added each score to zeros: the vector must be the periodic
    # continuation of its first `distinct` entries, bit for bit, whatever the number of ranks
    periodic = bool(np.array_equal(sc, np.resize(sc[:distinct], total)))
    return {"config": "configs[3] / SURVEY 8d config 4: fixed stream, contiguous shards, one reduce of scores to rank 0",
            "total_pairs": total, "seconds_first_submit_to_scores_on_rank0": dt, "value": total / dt, "unit": "frame-pairs/s",
            "scaling": "strong", "ranks_seen": ctx.dist.get_world_size() if ctx.dist is not None else 1,
            "pairs_per_rank": -(-total // ctx.world), "batch": B,
            "scores_periodic_bit_identical": periodic, "scores_sha256_16": hashlib.sha256(sc.tobytes()).hexdigest()[:16],
            "score_mean": float(sc.mean())}


def packed10_pairs(ctx, w, h, distinct):
    """`distinct` 10-bit pairs as page-locked PACKED pictures (tm_engine_set_frame_i420p10: three samples per 32-bit word): Y, Cb, Cr
    word planes back to back in one tensor per frame, like the CLI's ring slots.  Returns [(ref planes, dis planes)], bytes per pair."""
    tm, np, torch = ctx.tm, ctx.np, ctx.torch
    cw, ch = (w + 1) // 2, (h + 1) // 2
    wy, wc = tm.synth.p10_row_words(w), tm.synth.p10_row_words(cw)
    out = []
    for n in range(distinct):
        sides = []
        for planes in tm.synth.yuv420_pair(w, h, n, 10):
            flat = np.concatenate([tm.synth.p10_pack_plane(p).reshape(-1) for p in planes])
            t = torch.from_numpy(flat.view(np.int32)).pin_memory()
            sides.append((t[:h * wy].view(h, wy), t[h * wy:h * wy + ch * wc].view(ch, wc), t[h * wy + ch * wc:].view(ch, wc)))
        out.append(tuple(sides))
    return out, 2 * (h * wy + 2 * ch * wc) * 4


def run_host_fed(ctx, args, name, B, steps, warmup, packed10=False):
    """every step uploads its B pairs from page-locked host memory (TM_MEM_HOST_PINNED: asynchronous
    DMA on the engine's stream).  Two engines ping-pong, so the upload of batch k+1 overlaps the kernels of batch k -- the
    arrangement of the CLI's compute_all.  PCIe-inclusive: reported beside `value`, never as `value`.
    packed10: the same 10-bit pictures handed over as tm_engine_set_frame_i420p10 frames (10.7 instead of 16 bits per sample on the link)."""
    tm, torch = ctx.tm, ctx.torch
    w, h, kind, _, _ = WORKLOADS[name]
    distinct = max(1, min(args.distinct if w * h <= 1920 * 1080 else 2, B))
    p10 = packed10_pairs(ctx, w, h, distinct) if packed10 else None
    tm.set_placement_candidates(1)
    engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B) for _ in range(2)]
    tm.set_placement_candidates(8)
    busy = [False, False]
    n_scores = 0

    def submit(k):
        nonlocal n_scores
        e = engs[k & 1]
        if busy[k & 1]:
            e.sync()
            n_scores += len(e.scores_batch(B))
        if p10 is not None:
            for slot in range(B):
                fr, fd = p10[0][(k * B + slot) % distinct]
                e.set_pair(slot, tm.HwFrame.i420p10(*fr), tm.HwFrame.i420p10(*fd))
        else:
            ctx.fill_slots(e, name, distinct, k * B, B, pinned=True)
        e.compute_async(B)
        busy[k & 1] = True

    for k in range(warmup + 2):
        submit(k)
    for e in engs:
        e.sync()
    busy[:] = [False, False]
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        submit(k)
    for i, e in enumerate(engs):
        if busy[i]:
            e.sync()
            n_scores += len(e.scores_batch(B))
    torch.cuda.synchronize()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    for e in engs:
        e.close()
    in_bytes = p10[1] if p10 is not None else w * h * 3 // 2 * (1 if kind == "nv12" else 2) * 2
    return {"config": "SURVEY 8d config 2 (ii): pinned host -> H2D every pair, two engines ping-pong" + (", 10-bit samples packed three to a word (tm_engine_set_frame_i420p10)" if packed10 else ""), "workload": name,
            "value": ctx.world * B * steps / dt, "unit": "frame-pairs/s", "ms_per_step": dt / steps * 1e3, "pairs_per_step_per_gpu": B,
            "h2d_GBs_per_gpu": B * steps * in_bytes / dt / 1e9, "note": "PCIe-inclusive; includes the Python loop's 2 x B set_frame calls per step"}


def run_batch_curve(ctx, args, name, head_B, head_res):
    """the headline workload at the reference's own call granularity and up: compute_one is ONE pair per call
    (crates/turbo-metrics/src/lib.rs:268-360).  Per batch size: a fresh engine, ~60 ms of settling, ~0.25 s timed."""
    import subprocess
    tm, torch = ctx.tm, ctx.torch
    w, h, kind, _, _ = WORKLOADS[name]
    out = []
    for B in BATCH_CURVE:
        if B == head_B:
            out.append({"batch": B, "value": head_res["value"], "ms_per_step": head_res["ms_per_step"], "engine_mem_GB": head_res["config"]["engine_mem_GB"], "from": "headline run"})
            continue
        eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
        ctx.fill_slots(eng, name, max(1, min(args.distinct, B)), 0, B)
        t0 = time.perf_counter()
        n0 = 0
        while time.perf_counter() - t0 < 0.06:
            eng.compute_async(B); eng.sync(); n0 += 1
        k = max(10, int(0.25 / ((time.perf_counter() - t0) / n0)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            eng.compute_async(B)
            eng.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out.append({"batch": B, "value": B * k / dt, "ms_per_step": dt / k * 1e3, "steps": k, "engine_mem_GB": round(eng.mem_usage() / 1e9, 2)})
        if B <= 8:  # the same call granularity with several launches in flight: n engines take turns
            # (engines of their own, created together: the runtime spreads streams over its four hardware queues in the order in which
            # they are created, and an engine whose stream shares a queue with another's cannot run beside it)
            ref_scores = [sc.ssimulacra2 for sc in eng.scores_batch(B)]
            eng.close()
            engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B) for _ in range(2)]
            for e in engs:
                ctx.fill_slots(e, name, max(1, min(args.distinct, B)), 0, B)
            eng, extra = engs[0], engs[1:]
            fl = {}
            for n in (2,):  # (four in flight: 8.4 k pairs/s at one pair per launch when every engine's stream gets a hardware queue of its own --
                # the runtime has four by default and hands them out by its own rules: tools/inflight_probe.py)
                def turns(steps):
                    busy = [False] * n
                    for i in range(steps):
                        e = engs[i % n]
                        if busy[i % n]:
                            e.sync()
                        e.compute_async(B); busy[i % n] = True
                    for i in range(n):
                        if busy[i]:
                            engs[i].sync()
                turns(max(8, k // 10))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                turns(k)
                torch.cuda.synchronize()
                fl[str(n)] = B * k / (time.perf_counter() - t0)
            same = all([sc.ssimulacra2 for sc in e.scores_batch(B)] == ref_scores for e in engs)
            if B == 1:
                # the same through the host layer's API for reference-style callers: compute_one_deferred(pair) -> ticket, collect(ticket)
                # one call later (set_pair + launch per call, the scores of every pair fetched: what replaces compute_one in the
                # reference's loop, turbo-metrics-cli/src/main.rs:290-326)
                for e in extra:
                    e.close()
                extra = []
                kind = WORKLOADS[name][2]
                mk = tm.HwFrame.nv12 if kind == "nv12" else tm.HwFrame.p016
                sf = ctx.surfaces(name, max(1, min(args.distinct, 8)))
                pairs_hw = [(mk(rt, rp, rch), mk(dt_, dp, dch)) for (rt, rp, rch), (dt_, dp, dch) in sf]
                blocking = [eng.compute_one(a, b).ssimulacra2 for a, b in pairs_hw]

                def deferred(steps, depth):
                    got, tickets = [], []
                    for i in range(steps):
                        tickets.append(eng.compute_one_deferred(*pairs_hw[i % len(pairs_hw)]))
                        if len(tickets) >= depth:  # pair k is collected after pair k + depth - 1 went in
                            got.append(eng.collect(tickets.pop(0)).ssimulacra2)
                    got.extend(eng.collect(t).ssimulacra2 for t in tickets)
                    return got
                deferred(max(8, k // 10), 2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = deferred(k, 2)
                fl["2_api"] = k / (time.perf_counter() - t0)
                same = same and got[:len(blocking)] == blocking
                # set_deferred_depth(4): four engines taking turns.  In a process of its own, like a caller's: the runtime hands its four
                # hardware queues out by the order in which streams were created and freed, and this process has created and freed
                # dozens of engines by now (in-process the same loop measured 4.9 k, a fresh process 8.3 k: profiles/r06y5_deferred_depth.log)
                if name == "1080p_nv12":
                    try:
                        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "deferred_depth_probe.py"), "--one", "4", str(w), str(h)],
                                           capture_output=True, text=True, timeout=300)
                        fl["4_api"] = float(json.loads(r.stdout.strip().splitlines()[-1])["pairs_per_s"])
                    except Exception as ex:  # noqa: BLE001 -- a leg of the side record, never the line
                        out[-1]["in_flight_4_api_error"] = repr(ex)[:200]
            out[-1]["in_flight"] = fl
            out[-1]["in_flight_scores_identical"] = same
            for e in extra:
                e.close()
        eng.close()
    return {"workload": name, "note": "one compute_async + sync per step, inputs resident in HBM; batch 1 = the reference's compute_one granularity; "
            "in_flight: pairs/s of the same launches with 2 of them in flight (two engines taking turns)",
            "points": out}


def run_cli_end_to_end(ctx):
    """the C++ CLI (turbo-metrics_amd/bin/turbo-metrics) end to end on planar Y4M clips in tmpfs: file -> page-locked ring ->
    upload -> SSIMULACRA2 -> JSON lines on stdout; the figure is the CLI's own "Processed ... (N fps)" line (engine creation
    excluded, everything per frame included).  A child process; clips are written and removed here."""
    import re
    import subprocess
    np, tm = ctx.np, ctx.tm
    cli = os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")
    if not os.path.exists(cli):
        return {"error": "turbo-metrics binary not built"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    out = {}
    for tag, w, h, bits, frames in (("1080p_yuv420p", 1920, 1080, 8, 1536), ("4k_yuv420p10", 3840, 2160, 10, 192)):  # 4.8 GB per clip
        frame_bytes = w * h * 3 // 2 * (1 if bits == 8 else 2)
        paths = []
        try:
            st = os.statvfs(base)
            if st.f_bavail * st.f_frsize < 2 * frames * frame_bytes + (2 << 30):
                out[tag] = {"error": f"not enough room in {base}"}
                continue
            paths = [os.path.join(base, f"tm_bench_{os.getpid()}_{tag}_{s}.y4m") for s in ("ref", "dis")]
            pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(2)]
            for side, pth in enumerate(paths):
                with open(pth, "wb") as f:
                    f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 C420{'jpeg' if bits == 8 else 'p10'}\n".encode())
                    blobs = [b"FRAME\n" + b"".join(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes() for pl in pr[side]) for pr in pairs]
                    for i in range(frames):
                        f.write(blobs[i % len(blobs)])
            res = {"pairs": frames, "clip_GB_each": round(os.path.getsize(paths[0]) / 1e9, 2), "clips_in": base}
            # the first pass over a clip this process has JUST WRITTEN into tmpfs runs at a third of the later ones whatever its
            # arguments: the first read of just-written tmpfs pages does not scale with threads (12-14 GB/s with 1, 4 or 8 readers,
            # 58-65 on the second pass; tools/first_pass_probe.sh, profiles/r04y_first_pass_probe.log).  That is an artefact of
            # making the clip here, not what a user's single run over an existing file sees: it is kept in the detail record as
            # `tmpfs_just_written` and stays off the compact line.
            for label, extra in (("tmpfs_just_written", []), ("default", []), ("batch16", ["--batch", "16"]),
                                 # the reference's own loop -- one blocking compute_one per pair -- and the same loop on
                                 # compute_one_deferred + collect (two pairs in flight): what a reference-style caller gets end to end
                                 ("loop_reference", ["--loop", "reference"]), ("loop_deferred", ["--loop", "deferred"]),
                                 ("loop_deferred4", ["--loop", "deferred", "--in-flight", "4"])) + (
                                 # 10-bit clips: the readers pack three samples to a word on the way into the page-locked ring (default since
                                 # round 6); TM_PACK10=0 hands the 16-bit words over as before
                                 (("words16", ["TM_PACK10=0"]),) if bits == 10 else
                                 # the product's own multi-GPU arrangement with ONE rank: launcher, rank process, a one-rank RCCL communicator and the
                                 # ncclReduce of the score vector inside the CLI's clock (round 6; N > 1 needs N GPUs)
                                 (("ranks1_rccl", ["--ranks", "1", "TM_RANK_TRANSPORT=rccl", "TM_RANK_TIMEOUT_S=280"]),)):
                # `default` is the STEADY rate with the clip in the page cache: the passes right after the clip was written run at a
                # third to a half of the later ones, and for how many passes differs from box to box and run to run (1-2 of 5 in
                # tools-free probes: [2970, 4063, 7529, 7492, 7566]) -- the leg repeats until two consecutive passes agree within
                # 10 % (at most 6) and lists every pass
                passes = []
                for attempt in range(6 if label == "default" else 1):
                    t0 = time.perf_counter()
                    env_extra = dict(a.split("=", 1) for a in extra if "=" in a and not a.startswith("-"))
                    r = subprocess.run([cli, paths[0], paths[1], "-m", "ssimulacra2", "--output", "json-lines"] + [a for a in extra if a.startswith("-") or "=" not in a],
                                       capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_extra))
                    wall = time.perf_counter() - t0
                    m = re.search(r"Processed: (\d+) .*?\((\d+) fps\)", r.stderr)
                    passes.append(int(m.group(2)) if m else None)
                    if r.returncode != 0 or passes[-1] is None or (len(passes) >= 2 and passes[-2] and abs(passes[-1] - passes[-2]) <= 0.1 * passes[-2]):
                        break
                lines = sum(1 for l in r.stdout.splitlines() if l.startswith("{"))
                res[label] = {"rc": r.returncode, "pairs_per_s": passes[-1], "process_wall_s": round(wall, 2), "score_lines": lines,
                              "args": " ".join(extra) or "(CLI defaults)", **({"passes": passes} if len(passes) > 1 else {})}
            out[tag] = res
        except Exception as ex:  # the leg is informative: never take the headline down with it
            out[tag] = {"error": repr(ex)[:200]}
        finally:
            for pth in paths:
                if os.path.exists(pth):
                    os.remove(pth)
    out["note"] = "host-fed through the CLI (PCIe + file reads inclusive): reported beside `value`, never as `value`"
    return out


def effective_cpus():
    return launch_module().effective_cpus()


def cpu_baseline(tm, w, h, kind, n_pairs):
    """CPU baselines on this host, bounded to ~15-20 s in total (reported, never the target): the restated reference CPU path
    (oracle/tm_cpu_path.c == examples/cpu.rs, single-threaded per pair like the original) after the same YUV->linear conversion
    the GPU path applies.  The frame-level parallel loop lives in the oracle library (tmo_cpu_path_run: pthreads, one pair per
    worker at a time, every worker's buffers allocated and touched once before the clock starts) -- round 3 ran Python threads around
    ctypes calls that malloc'ed 200 MB per pair.  Points: 1 thread, and one thread per CPU the process may really use
    (effective_cpus(): the container's cgroup quota, 16 on this pool's boxes, not the 256 CPUs it can see -- more threads than the
    quota only get the group throttled: 64 threads 33 pairs/s, 256 threads 20); plus the GPU-arithmetic oracle on 1 thread."""
    from oracle import oracle as O
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    bits = 8 if kind == "nv12" else 16
    pairs = [gen(w, h, n) for n in range(2)]
    if n_pairs <= 0:
        n_pairs = 12 if w * h <= 1920 * 1080 else 3  # ~5 s on one core
    dt1, _ = O.cpu_path_run(pairs, w, h, bits, n_pairs, 1)
    one = n_pairs / dt1
    (rs, rp, rch), (ds, dp, dch) = pairs[0]
    lr = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, 0)
    ld = O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, bits, 0)
    t0 = time.perf_counter()
    for _ in range(2):
        O.ssimulacra2_from_linear(lr, ld)
    dt_gpu_arith = (time.perf_counter() - t0) / 2
    cpus = effective_cpus()
    per_worker = 31 * 4 * w * h  # 25 planes of workspace + two linear RGB images
    try:
        import psutil
        cap = max(1, int(psutil.virtual_memory().available * 0.5 / per_worker))
    except Exception:
        cap = 64
    points = [{"cores": 1, "value": one, "pairs": n_pairs, "seconds": dt1}]
    for threads in sorted({min(cpus, cap)}):
        if threads <= 1:
            continue
        k = 4 * threads  # four pairs per worker: ~0.5 s per pair
        dtp, _ = O.cpu_path_run(pairs, w, h, bits, k, threads)
        points.append({"cores": threads, "value": k / dtp, "pairs": k, "seconds": dtp})
    top = points[-1]
    return {"value": one, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n_pairs} {w}x{h} {kind} pairs, YUV->linear + restated reference CPU path (oracle/tm_cpu_path.c == examples/cpu.rs), 1 thread",
            "seconds": dt1, "host_cpus": os.cpu_count(), "usable_cpus": cpus,
            "all_cores": {"value": top["value"], "cores": top["cores"], "scaling_vs_1core": top["value"] / one, "efficiency": top["value"] / one / top["cores"],
                          "sample": f"{top['pairs']} pairs over {top['cores']} pthreads = the CPUs this container may use (cgroup quota / affinity; {os.cpu_count()} visible); tmo_cpu_path_run: one pair per worker at a time, buffers allocated once per worker",
                          "seconds": top["seconds"]},
            "points": points,
            "gpu_arithmetic_oracle_1thread_pairs_per_s": 1.0 / dt_gpu_arith}


LINE_LIMIT = 4096  # bytes: the driver reads the LAST stdout line; BENCH_r03's 22-KB line came back unparsed (tests/test_bench_line.py)


def _r(x, nd=4):
    """numbers on the compact line: 4 significant decimals are what the figures are good for"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return round(float(x), nd) if abs(x) < 1e6 else float(f"{x:.7g}")


def compact_line(d):
    """The ONE line the driver parses: the contract fields, `roofline`, `cpu_baseline` and a one-level `summary`, <= LINE_LIMIT
    bytes.  `d` is the full record (every leg with its kernels, members, stages ...), which goes to bench_detail.json."""
    cfg = d.get("config", {})
    out = {k: _r(d.get(k)) for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if d.get("per_rank"):
        out["per_rank"] = [_r(v, 1) for v in d["per_rank"]]  # [slowest, fastest] rank, pairs/s over its own K steps
        out["reduce_ms"] = _r(d.get("reduce_ms"), 3)
    out["config"] = {k: cfg.get(k) for k in ("workload", "baseline_config", "width", "height", "input", "pairs_per_step_per_gpu",
                                             "metrics", "full_sums", "pipeline_depth", "engine_mem_GB", "parallelism") if k in cfg}
    rf = d.get("roofline") or {}
    keep = ("bound", "kind", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
            "bytes_model", "frac_alone", "avg_launch_ms_alone", "longest_hbm_bound_member")
    out["roofline"] = {k: _r(rf.get(k)) for k in keep if k in rf}
    cmp_ = d.get("compare") or {}
    if cmp_.get("full_sums"):  # like for like with the reference (all 108 sums): SURVEY 8d's unrestricted byte model
        out["roofline"]["full_sums_frac"] = _r(cmp_.get("blur_reduce_stage_frac"))
        out["roofline"]["full_sums_value"] = _r(cmp_.get("value"), 1)
    if len(out["roofline"].get("bytes_model", "")) > 160:
        out["roofline"]["bytes_model"] = out["roofline"]["bytes_model"][:157] + "..."
    cb = d.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:150], "host_cpus": cb.get("host_cpus"), "usable_cpus": cb.get("usable_cpus")}
        if cb.get("all_cores"):
            out["cpu_baseline"]["all_cores"] = {k: _r(cb["all_cores"].get(k)) for k in ("value", "cores", "scaling_vs_1core", "efficiency") if k in cb["all_cores"]}
        if cb.get("points"):
            out["cpu_baseline"]["points"] = [[p["cores"], _r(p["value"], 2)] for p in cb["points"]]
    sm = {}
    for name, w in (d.get("workloads") or {}).items():
        sm[name] = [_r(w.get("value"), 1), _r((w.get("roofline") or {}).get("frac")), [_r(v, 3) for v in w.get("stage_ms") or []]]
    if d.get("stage_ms"):
        sm["stage_ms"] = [_r(v, 3) for v in d["stage_ms"]]
    if sm:
        sm["_workloads"] = "[pairs/s, blur+reduce frac, stage ms [ingest,col,row,edge(beside),ssim|finish]]"
    fs = d.get("fixed_stream")
    if fs:
        sm["fixed_stream"] = {"pairs": fs.get("total_pairs"), "value": _r(fs.get("value"), 1), "scaling": fs.get("scaling"),
                              "bit_identical": fs.get("scores_periodic_bit_identical"), "sha": fs.get("scores_sha256_16")}
        if fs.get("long"):
            sm["fixed_stream_long"] = {"pairs": fs["long"].get("total_pairs"), "value": _r(fs["long"].get("value"), 1),
                                       "bit_identical": fs["long"].get("scores_periodic_bit_identical"), "sha": fs["long"].get("scores_sha256_16")}
    hf = d.get("host_fed")
    if hf:
        sm["host_fed"] = {k: [_r(v.get("value"), 1), _r(v.get("h2d_GBs_per_gpu"), 1)] for k, v in hf.items()}
        sm["_host_fed"] = "[pairs/s, H2D GB/s] PCIe-inclusive, never `value`"
    bc = d.get("batch_curve")
    if bc:
        sm["batch_curve"] = [[p["batch"], _r(p["value"], 0)] for p in bc.get("points", [])]
        fl = [[p["batch"], n, _r(v, 0)] for p in bc.get("points", []) for n, v in sorted((p.get("in_flight") or {}).items())]
        if fl:
            sm["batch_curve_in_flight"] = fl
            sm["_batch_curve_in_flight"] = "[pairs per launch, launches in flight (2_api / 4_api: via compute_one_deferred/collect at that depth), pairs/s]"
    cli = d.get("cli_end_to_end")
    if cli:
        sm["cli_end_to_end"] = {tag: {lab: v[lab].get("pairs_per_s") for lab in ("default", "batch16", "loop_reference", "loop_deferred", "loop_deferred4", "words16", "ranks1_rccl") if isinstance(v.get(lab), dict)}
                                if isinstance(v, dict) and "error" not in v else (v.get("error", "")[:60] if isinstance(v, dict) else None)
                                for tag, v in cli.items() if tag != "note"}
    pl = d.get("pipeline")
    if pl:
        sm["pipeline"] = {k: _r(v) for k, v in pl.items() if not isinstance(v, (dict, list))}
    if d.get("leg_errors"):
        sm["leg_errors"] = {k: str(v)[:80] for k, v in list(d["leg_errors"].items())[:4]}
    if d.get("score_mean") is not None:
        sm["score_mean"] = _r(d["score_mean"], 6)
    sm["detail"] = d.get("detail_file")
    out["summary"] = sm
    line = json.dumps(out, separators=(",", ":"))
    # never let an unforeseen long string take the line over the limit: drop the least important entries first
    for victim in ("_batch_curve_in_flight", "batch_curve_in_flight", "cli_end_to_end", "batch_curve", "host_fed", "_host_fed", "_workloads", "pipeline", "fixed_stream_long"):
        if len(line) <= LINE_LIMIT:
            break
        sm.pop(victim, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:
        out.pop("summary", None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def write_detail(d, path=None):
    """the full record: bench_detail.json beside bench.py (and under gpurun_out/ when that exists, so that it comes back from a GPU box)"""
    paths = [path] if path else [os.path.join(ROOT, "bench_detail.json")]
    if not path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(d, f, indent=1)
            written = written or os.path.relpath(pth, ROOT)
        except OSError:
            pass
    return written


def run_rank(args):
    ctx = Ctx(args)
    head_name = args.workload or "1080p_nv12"
    mets = set(args.metrics.split(","))
    B = args.batch or (SHARED_DEVICE_BATCH[head_name] if ctx.shared_device else WORKLOADS[head_name][3])
    # the other workloads, the host-fed leg, the batch curve and the CLI describe ONE GPU: with several ranks the line carries the weak
    # leg (`value`) and the strong leg (`fixed_stream`) only -- eight ranks generating 4K frames on a shared CPU quota is setup time that
    # measures nothing
    extras = args.workload is None and not args.no_extras and mets == {"ssimulacra2"} and ctx.world == 1
    res, eng, distinct = run_workload(ctx, args, head_name, mets, B, args.steps, args.warmup, args.settle_ms,
                                      compare=not args.no_compare, keep_engine=True)
    fixed = None
    if "ssimulacra2" in mets and (not args.no_extras or args.stream_pairs is not None):
        eng.set_full_sums(args.full_sums)
        fixed = run_fixed_stream(ctx, eng, head_name, B, distinct, args.stream_pairs or STREAM_PAIRS)
        if args.stream_pairs is None:  # the default line: a second, longer stream (>= 0.15 s of kernels per rank at 8 GPUs)
            longer = run_fixed_stream(ctx, eng, head_name, B, distinct, STREAM_PAIRS_LONG)
            if fixed is not None and longer is not None:
                fixed["long"] = {k: longer[k] for k in ("total_pairs", "seconds_first_submit_to_scores_on_rank0", "value", "pairs_per_rank", "scores_periodic_bit_identical", "scores_sha256_16")}
    # The legs below describe the machine around the headline (other workloads, PCIe, call granularity, the CLI): a failure in one of
    # them -- a box out of memory, no room in tmpfs -- is recorded on the line and never takes the headline down with it.
    workloads, host_fed, leg_errors = {}, None, {}

    def leg(name, fn):
        try:
            return fn()
        except Exception as ex:  # noqa: BLE001
            leg_errors[name] = repr(ex)[:200]
            try:
                ctx.torch.cuda.synchronize()
            except Exception:
                pass
            return None

    # The CLI leg runs FIRST, while the headline engine is still alive.  Behind the other legs -- four engines of 31 ... 46 GB created and
    # closed by THIS process, the host-fed engines, the batch curve's -- the CLI child ran at half its rate for four passes and more
    # (3.9 k instead of 7.8 k pairs/s at 1080p); with one engine closed before it, for one or two passes; with nothing closed before it,
    # from the second pass on it is at its rate (three runs: 7.2-7.9 k).  The mechanism is not established: inside one process neither
    # the device's copy rate nor its H2D rate moves after freeing 46 or 150 GB (profiles/r05u_cli_after_frees.log holds both probes).
    cli = None
    if extras and ctx.world == 1 and head_name == "1080p_nv12" and not args.no_cli:
        cli = leg("cli_end_to_end", lambda: run_cli_end_to_end(ctx))
    eng.close()
    if extras:
        # the same K, W and settling as the headline: a 10-step run after 150 ms of settling differed by 13 % from box to box (r04)
        ks, kw = max(2, args.steps), args.warmup
        for wl, m in EXTRAS:
            tag = wl + ("_fused" if m == FUSED else "")
            got = leg(tag, lambda: run_workload(ctx, args, wl, set(m.split(",")), WORKLOADS[wl][3], ks, kw, args.settle_ms, compare=False))
            if got is not None:
                r = got[0]
                r.pop("unit", None)
                workloads[tag] = r
        if ctx.world == 1:  # a per-GPU PCIe figure: measured on one GPU only (>= 40 steps: 8 pairs x 10 steps was too short to be stable)
            host_fed = {}
            for wl in ("1080p_nv12", "4k_p016"):
                got = leg("host_fed_" + wl, lambda: run_host_fed(ctx, args, wl, 32 if wl == "1080p_nv12" else 8, max(ks, 40), min(kw, 2)))
                if got is not None:
                    host_fed[wl] = got
            got = leg("host_fed_4k_p10", lambda: run_host_fed(ctx, args, "4k_p016", 8, max(ks, 40), min(kw, 2), packed10=True))
            if got is not None:
                host_fed["4k_p10"] = got
    batch_curve = None
    if extras and ctx.world == 1 and head_name == "1080p_nv12":
        batch_curve = leg("batch_curve", lambda: run_batch_curve(ctx, args, head_name, B, res))
    if ctx.rank == 0:
        out = {
            "metric": "ssimulacra2_frame_pairs_per_sec" if "ssimulacra2" in mets else "frame_pairs_per_sec",
            "value": res["value"],
            "unit": "frame-pairs/s",
            "n_gpus": ctx.world,
            "ranks_seen": ctx.dist.get_world_size() if ctx.dist is not None else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
        }
        for k in ("config", "roofline", "kernels", "kernels_alone", "stages", "score_mean", "compare", "per_rank", "reduce_ms", "reduce_first_ms", "stage_ms", "rank_cpu", "device_mem_used_GB"):
            if k in res:
                out[k] = res[k]
        if leg_errors:
            out["leg_errors"] = leg_errors
        if workloads:
            out["workloads"] = workloads
        if fixed is not None:
            out["fixed_stream"] = fixed
        if host_fed:
            out["host_fed"] = host_fed
        if batch_curve:
            out["batch_curve"] = batch_curve
        if cli:
            out["cli_end_to_end"] = cli
        if ctx.world == 1 and not args.no_cpu_baseline:
            w, h, kind = WORKLOADS[head_name][:3]
            out["cpu_baseline"] = cpu_baseline(ctx.tm, w, h, kind, args.cpu_pairs)
        out["detail_file"] = args.detail_file or "bench_detail.json"
        write_detail(out, args.detail_file)
        print(compact_line(out), flush=True)
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


def main():
    args = parse_args()
    # dmabuf IPC is what RCCL (and tensor sharing between processes) needs on this driver; the pool exports it, a bare shell may not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher.  Nothing has touched the GPU (torch is not even imported yet).
        sys.exit(launch_module().spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
