"""N>1 path on CPU: two gloo ranks shard a stream of frame pairs, score their own block and reduce the
zero-padded score vector to rank 0 -- the same code (`tm.shard`) bench.py runs over RCCL.  The per-frame
"engine" here is the CPU oracle on tiny frames (tests may use the oracle as a stand-in; there is no GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _score_frame(n, w=40, h=24):
    from oracle import oracle as O
    from tm_pkg import tm
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    lr = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 8, 0)
    ld = O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 8, 0)
    s, _ = O.ssimulacra2_from_linear(lr, ld)
    _, p = O.psnr(lr, ld)
    return [s, p]


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tm_pkg import tm
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = tm.shard.shard_range(n_frames, rank, world)
    local = [_score_frame(n) for n in range(lo, hi)]
    out = tm.shard.reduce_scores(local, lo, n_frames, n_metrics=2, dist=dist)
    span = tm.shard.min_max_over_ranks(100.0 + 7.0 * rank, dist)  # bench.py's per_rank: [slowest, fastest]
    dist.barrier()
    if rank == 0:
        q.put((out, span))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 1])
def test_two_rank_shard_and_reduce_matches_single_process(n_frames):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, span = q.get(timeout=120)
    assert span == [100.0, 107.0]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([_score_frame(n) for n in range(n_frames)])
    assert got.shape == (n_frames, 2)
    assert np.array_equal(got, want)  # bit-identical: every score is added to zeros exactly once


def test_shard_ranges_cover_the_stream():
    sys.path.insert(0, ROOT)
    from tm_pkg import tm
    assert tm.shard.min_max_over_ranks(3.5) == [3.5, 3.5]  # no process group: one rank
    for n in (0, 1, 7, 8, 2048):
        for world in (1, 2, 4, 8):
            blocks = [tm.shard.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert all(hi - lo <= -(-n // world) for lo, hi in blocks)
