"""GPU tier: bench.py's multi-rank path really starts N ranks.  On the 1-GPU test box the two ranks share
the device and the one collective goes through gloo (TM_BENCH_BACKEND=gloo); the driver's 8-GPU run uses RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--workload", "1080p_nv12", "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--no-cpu-baseline", "--no-compare",
          "--batch", "8", "--stream-pairs", "72"]


def _run(extra, common, env_extra=None):
    """bench.py as the driver runs it: ONE compact JSON line on stdout (<= 4 KB, the contract fields + roofline + summary);
    returns the full record (bench_detail.json, --detail-file) after checking the line against it"""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + common + ["--detail-file", detail],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
        assert len(lines[0]) <= 4096, len(lines[0])
        line = json.loads(lines[0])
        full = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "scaling", "dtype", "data", "config", "roofline", "summary"):
        assert k in line, k
    assert line["n_gpus"] == full["n_gpus"] and abs(line["value"] - full["value"]) < 1e-3 * full["value"]
    assert line["roofline"]["kind"] in ("group", "kernel") and abs(line["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    if "fixed_stream" in full:
        assert line["summary"]["fixed_stream"]["sha"] == full["fixed_stream"]["scores_sha256_16"]
    full["_line"] = line
    return full


def _bench(extra, env_extra=None):
    return _run(extra, COMMON, env_extra)


def test_two_ranks_are_really_started_and_reproduce_the_one_rank_scores():
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2"], {"TM_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["fixed_stream"]["ranks_seen"] == 2
    assert two["scaling"] == "weak" and two["fixed_stream"]["scaling"] == "strong"
    # weak leg: every rank ran its own 8 pairs per step
    assert two["config"]["pairs_per_step_per_gpu"] == 8 and two["value"] > 0
    # strong leg: 72 pairs = 36 per rank (4 full batches + a short one), contiguous shards, one reduce: the same 72 scores, bit for bit
    assert two["fixed_stream"]["total_pairs"] == 72 and two["fixed_stream"]["pairs_per_rank"] == 36
    assert one["fixed_stream"]["scores_periodic_bit_identical"] and two["fixed_stream"]["scores_periodic_bit_identical"]
    assert one["fixed_stream"]["scores_sha256_16"] == two["fixed_stream"]["scores_sha256_16"]
    for k in ("roofline", "kernels", "stages"):
        assert k in two


def test_eight_ranks_rendezvous_shard_and_reduce_like_one_rank():
    """the driver's widest run, without eight GPUs: 8 rank processes on one device (gloo carries the one collective), a fixed
    stream of 2 048 pairs = 256 per rank in 128 launches of 2, one reduce to rank 0 -- the same 2 048 scores as one rank, bit
    for bit; the whole thing well inside two minutes"""
    import time
    args = ["--workload", "1080p_nv12", "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--no-cpu-baseline", "--no-compare",
            "--batch", "2", "--stream-pairs", "2048", "--no-extras"]

    def run(extra, env_extra=None):
        t0 = time.time()
        return _run(extra, args, env_extra), time.time() - t0

    one, _ = run(["--gpus", "1"])
    eight, wall = run(["--gpus", "8"], {"TM_BENCH_BACKEND": "gloo"})
    assert eight["n_gpus"] == 8 and eight["ranks_seen"] == 8 and eight["fixed_stream"]["ranks_seen"] == 8
    assert eight["fixed_stream"]["total_pairs"] == 2048 and eight["fixed_stream"]["pairs_per_rank"] == 256
    assert eight["fixed_stream"]["scores_periodic_bit_identical"]
    assert eight["fixed_stream"]["scores_sha256_16"] == one["fixed_stream"]["scores_sha256_16"]
    assert "workloads" not in eight and "batch_curve" not in eight  # --no-extras
    assert "gloo" in eight["_line"]["config"]["parallelism"]  # the line states the backend that carried the reduce
    assert eight["_line"]["summary"]["fixed_stream"]["pairs"] == 2048
    assert wall < 120.0, wall


def test_the_drivers_eight_gpu_command_with_default_arguments_on_one_device():
    """the command the driver runs at round end -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps 20 --warmup 5`, batch 64, both fixed streams -- rehearsed
    on ONE device (TM_BENCH_BACKEND=gloo: the ranks share device 0, gloo carries the collectives).  Every rank caps its thread
    pools to its share of the CPU quota and binds next to its device before any work; the line names the real backend, shows
    the slowest and the fastest rank and the time of the one reduce; < 600 s, <= 4 KB.  The record goes to gpurun_out/ (copied
    to profiles/r05_bench_8ranks_gloo_one_device.json)."""
    import tempfile
    import time
    from tm_pkg import tm
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update({"TM_BENCH_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(tm.launch.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
               "--detail-file", detail]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1 and len(lines[0]) <= 4096, (len(lines), [len(l) for l in lines])
        line = json.loads(lines[0])
        full = json.load(open(detail))
    assert wall < 600.0, wall
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["steps"] == 20 and line["warmup"] == 5
    assert line["config"]["pairs_per_step_per_gpu"] == 64 and "gloo" in line["config"]["parallelism"]
    lo, hi = line["per_rank"]
    assert 0 < lo <= hi and line["reduce_ms"] >= 0
    # value = the pairs of all ranks over the slowest rank's time: never above the sum of the per-rank rates
    assert line["value"] <= 8 * hi * 1.001
    sm = line["summary"]
    assert sm["fixed_stream"]["pairs"] == 2048 and sm["fixed_stream"]["bit_identical"]
    assert sm["fixed_stream_long"]["pairs"] == 16384 and sm["fixed_stream_long"]["bit_identical"]
    assert full["rank_cpu"]["threads"] >= 1 and full["rank_cpu"]["threads"] <= max(1, tm.launch.effective_cpus() // 8)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r05_bench_8ranks_gloo_one_device.json"), "w") as f:
            json.dump({"command": " ".join(cmd[1:]).replace(ROOT + "/", ""), "env": {"TM_BENCH_BACKEND": "gloo"}, "wall_s": round(wall, 1),
                       "line_bytes": len(lines[0]), "device_mem_used_GB": full.get("device_mem_used_GB"), "rank_cpu": full["rank_cpu"],
                       "line": line}, f, indent=1)


def test_asking_for_more_gpus_than_exist_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TM_BENCH_BACKEND")}
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + COMMON, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "device(s) visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_one_rank_process_group_runs_every_collective_through_rccl():
    """the RCCL branch on hardware: a process group of one rank (torchrun-style variables, backend nccl = RCCL): init with
    device_id, barrier, all_reduce(MAX) of the timing, reduce(SUM) of the scores on device tensors, destroy -- and the same
    numbers as without a process group"""
    from tm_pkg import tm
    free_port = tm.launch.free_port
    plain = _bench(["--gpus", "1"])
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
           "TM_BENCH_FORCE_DIST": "1", "TM_BENCH_BACKEND": "nccl", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    grp = _bench(["--gpus", "1"], env)
    assert grp["n_gpus"] == 1 and grp["ranks_seen"] == 1 and "RCCL" in grp["config"]["parallelism"]
    assert grp["fixed_stream"]["scores_sha256_16"] == plain["fixed_stream"]["scores_sha256_16"]
    assert grp["score_mean"] == plain["score_mean"]
