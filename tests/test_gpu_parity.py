"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.
Bar: bit-exact for every f32 plane and the integer SSE; 1e-12 relative for the f64 sums (summation
order differs); |score difference| <= 1e-4 per north_star (observed ~1e-12)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tm_pkg import tm

pytestmark = pytest.mark.gpu
F = tm.ffi


@pytest.fixture(scope="module", autouse=True)
def _init():
    tm.init_hip(0)


def oracle_linear(f, w, h):
    if f.kind in ("nv12", "p016"):
        return O.yuv420_biplanar_to_linear(np.asarray(f.data), f.pitch, f.coded_height, w, h, 8 if f.kind == "nv12" else 16, int(f.matrix))
    fn = {"rgb8": O.rgb8_to_linear, "rgb16": O.rgb16_to_linear, "rgbf32": O.rgbf32_to_linear, "linear_f32": O.linear_packed_to_planar}[f.kind]
    return fn(np.asarray(f.data))


def nv12_frames(w, h, n, matrix=tm.ColorMatrix.BT709):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    return tm.HwFrame.nv12(rs, rp, rch, matrix), tm.HwFrame.nv12(ds, dp, dch, matrix)


def p016_frames(w, h, n):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, n)
    return tm.HwFrame.p016(rs, rp, rch), tm.HwFrame.p016(ds, dp, dch)


def check_planes(eng, slot, fr, fd, w, h, scales=range(6), have_linear=False, have_xybt=False):
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    sums, pyr = O.ssimulacra2_sums(lin[0], lin[1], want_xyb=True)
    for side in range(2):
        for c in range(3):
            if have_linear:  # only the reference pipeline keeps linear RGB in HBM
                assert np.array_equal(eng.read_plane(slot, F.TM_PLANE_LINEAR, 0, side, c), lin[side][c]), ("linear", side, c)
    for s in scales:
        for side in range(2):
            for c in range(3):
                assert np.array_equal(eng.read_plane(slot, F.TM_PLANE_XYB, s, side, c), pyr[s][side][c]), ("xyb", s, side, c)
                if have_xybt:
                    assert np.array_equal(eng.read_plane(slot, F.TM_PLANE_XYB_T, s, side, c), pyr[s][side][c].T), ("xybt", s, side, c)
        _, cap = O.process_scale(pyr[s][0], pyr[s][1], capture=True)
        for p in range(5):
            for c in range(3):
                assert np.array_equal(eng.read_plane(slot, F.TM_PLANE_PASS1_T, s, p, c), cap["pass1"][p][c]), ("pass1", s, p, c)
    return lin, sums


def check_scores(eng, slot, lin, sums, w, h):
    np.testing.assert_allclose(eng.raw_sums(slot), sums, rtol=1e-12, atol=1e-300)
    want = O.score_from_sums(sums, w, h)
    got = eng.scores(slot)
    assert abs(got.ssimulacra2 - want) <= 1e-4  # north_star tolerance
    assert abs(got.ssimulacra2 - want) <= 1e-9  # what the design actually delivers
    if got.psnr is not None:
        sse, psnr = O.psnr(lin[0], lin[1])
        assert eng.sse(slot) == sse
        assert got.psnr == psnr  # bit-exact


@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (64, 64), (1, 1), (2, 5), (129, 20), (257, 131)])
def test_nv12_planes_and_scores_match_oracle(w, h):
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=3, full_sums=True)
    frames = [nv12_frames(w, h, n, tm.ColorMatrix(n % 3)) for n in range(3)]
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    eng.compute_async(3)
    eng.sync()
    for slot, (fr, fd) in enumerate(frames):
        lin, sums = check_planes(eng, slot, fr, fd, w, h)
        check_scores(eng, slot, lin, sums, w, h)
    eng.close()


@pytest.mark.parametrize("variant", [F.TM_VARIANT_REFERENCE, F.TM_VARIANT_WIDE_ROWS])
def test_reference_pipeline_and_wide_row_pass_are_bit_identical_with_the_default(variant):
    """TM_VARIANT_REFERENCE: the straight-line kernels kept as the on-device cross-check (they also keep linear RGB and a
    transposed XYB copy in HBM); TM_VARIANT_WIDE_ROWS: the row-pass instantiation of frames wider than 2560 pixels.  Both must
    reproduce the oracle bit for bit, like the default pipeline (test_nv12_planes_and_scores_match_oracle)."""
    w, h = 333, 203
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
    eng.set_variant(variant)
    frames = [nv12_frames(w, h, n) for n in range(2)]
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    eng.compute_async()
    eng.sync()
    ref = variant == F.TM_VARIANT_REFERENCE
    for slot, (fr, fd) in enumerate(frames):
        lin, sums = check_planes(eng, slot, fr, fd, w, h, have_linear=ref, have_xybt=ref)
        check_scores(eng, slot, lin, sums, w, h)
    got = [eng.raw_sums(i).copy() for i in range(2)]
    eng.set_variant(F.TM_VARIANT_DEFAULT)
    eng.compute_async()
    eng.sync()
    for i in range(2):
        np.testing.assert_allclose(eng.raw_sums(i), got[i], rtol=1e-12, atol=1e-300)
    with pytest.raises(tm.TmError):
        eng.set_variant(2)
    eng.close()
    with pytest.raises(tm.TmError):  # the reference pipeline has no SSIM / MS-SSIM stage
        e2 = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, ssim=True), batch=1)
        try:
            e2.set_variant(F.TM_VARIANT_REFERENCE)
        finally:
            e2.close()


@pytest.mark.parametrize("kind,w,h", [("nv12", 333, 203), ("p016", 258, 131), ("i420_10", 129, 67), ("nv12", 2, 5), ("nv12", 1921, 1079)])
def test_row_walking_ingest_equals_the_tile_ingest_and_the_oracle(kind, w, h):
    """k_ingest_rows (what a launch of one 4:2:0 kind runs: a lane = one quad of BOTH frames, a wave = 64 quads x N quad rows)
    against the oracle and against k_ingest_wave (TM_VARIANT_TILE_INGEST), for several N: every XYB plane of every scale, the
    integer SSE, the SSIM sums (they read the u8 planes), the 108 sums -- bit for bit between the two kernels."""
    want_ssim = min(w, h) >= 11
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=want_ssim), batch=2, full_sums=True)
    if kind == "nv12":
        frames = [nv12_frames(w, h, n, tm.ColorMatrix(n % 3)) for n in range(2)]
    elif kind == "p016":
        frames = [p016_frames(w, h, n) for n in range(2)]
    else:
        frames = []
        for n in range(2):
            ref, dis = tm.synth.yuv420_pair(w, h, n + 5, 10)
            frames.append(tuple(tm.HwFrame.i420(*(p.astype(np.uint16) for p in side), bits=10) for side in (ref, dis)))
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)

    def snapshot():
        eng.compute_async()
        eng.sync()
        planes = [eng.read_plane(slot, F.TM_PLANE_XYB, s, side, c).copy() for slot in range(2) for s in range(6) for side in range(2) for c in range(3)]
        return planes, [eng.raw_sums(i).copy() for i in range(2)], [eng.sse(i) for i in range(2)], [eng.ssim_sums(i).copy() for i in range(2)] if want_ssim else None

    eng.set_variant(F.TM_VARIANT_TILE_INGEST)
    tile = snapshot()
    eng.set_variant(F.TM_VARIANT_DEFAULT)
    # rows 0 (chosen per launch), 4, 8: pyramid levels 2..5 in the kernel's own epilogue (round 6); other counts and TM_VARIANT_UPPER_KERNEL: by
    # k_ingest_upper_rd, like the tile kernel
    for rows, variant in ((0, 0), (2, 0), (4, 0), (6, 0), (8, 0), (32, 0), (0, F.TM_VARIANT_UPPER_KERNEL), (8, F.TM_VARIANT_UPPER_KERNEL)):
        eng.set_variant(variant)
        assert F.lib().tm_engine_debug_set_ingest_rows(eng._h, rows) == 0
        got = snapshot()
        assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got[0], tile[0])), rows
        assert all(np.array_equal(a, b) for a, b in zip(got[1], tile[1])) and got[2] == tile[2], rows
        if want_ssim:
            assert all(np.array_equal(a, b) for a, b in zip(got[3], tile[3])), rows
    assert F.lib().tm_engine_debug_set_ingest_rows(eng._h, 3) != 0  # odd
    eng.set_variant(F.TM_VARIANT_DEFAULT)
    assert F.lib().tm_engine_debug_set_ingest_rows(eng._h, 0) == 0
    eng.compute_async(); eng.sync()
    if kind != "i420_10":  # (the planar kinds are checked against the repacked surface elsewhere)
        for slot, (fr, fd) in enumerate(frames):
            lin, sums = check_planes(eng, slot, fr, fd, w, h, scales=range(3))
            check_scores(eng, slot, lin, sums, w, h)
    eng.close()


@pytest.mark.parametrize("w,h,batch", [(333, 203, 2), (1920, 1080, 1), (70, 38, 3), (1920, 1080, 4)])
def test_multi_wave_row_pass_is_bit_identical_with_the_one_wave_row_pass(w, h, batch):
    """k_blur_h_jobs_split (what launches of up to ~2 600 row blocks run by default: eight waves per row block) against
    k_blur_h_jobs_x: the same 108 sums, bit for bit, pruned and full job tables, and the oracle's"""
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=batch)
    frames = [nv12_frames(w, h, n) for n in range(batch)]
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    for full in (False, True):
        eng.set_full_sums(full)
        got = {}
        for name, variant in (("split", F.TM_VARIANT_SPLIT_ROWS), ("whole", F.TM_VARIANT_WHOLE_ROWS), ("default", F.TM_VARIANT_DEFAULT)):
            eng.set_variant(variant)
            eng.compute_async()
            eng.sync()
            got[name] = [eng.raw_sums(i).copy() for i in range(batch)]
        for i in range(batch):
            assert np.array_equal(got["split"][i], got["whole"][i]) and np.array_equal(got["default"][i], got["whole"][i])
    eng.debug_set_param(F.TM_DBG_SPLIT_ROWS_BELOW, 0)  # the threshold is a tuning value: results do not depend on it
    eng.set_variant(F.TM_VARIANT_DEFAULT)
    eng.compute_async(); eng.sync()
    assert all(np.array_equal(eng.raw_sums(i), got["whole"][i]) for i in range(batch))
    # the column pass of a launch this small runs every role-wave as a workgroup of its own (one 1080p pair: 745 role-waves <= 800);
    # forbidden and forced, the pass-1 planes and the sums are the same bits
    planes = {}
    for name, below in (("five_wave_workgroups", 0), ("solo", 1 << 40)):
        eng.debug_set_param(F.TM_DBG_SOLO_COL_BELOW, below)
        eng.compute_async(); eng.sync()
        assert all(np.array_equal(eng.raw_sums(i), got["whole"][i]) for i in range(batch)), name
        planes[name] = [eng.read_plane(batch - 1, F.TM_PLANE_PASS1_T, s_, p_, 1) for s_ in (0, 2) for p_ in range(5)]
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(planes["solo"], planes["five_wave_workgroups"]))
    eng.debug_set_param(F.TM_DBG_SOLO_COL_BELOW, 800)
    if w * h <= 333 * 203:
        eng.set_variant(F.TM_VARIANT_SPLIT_ROWS)
        eng.compute_async()
        eng.sync()
        for slot, (fr, fd) in enumerate(frames):
            lin, sums = check_planes(eng, slot, fr, fd, w, h, scales=range(2))
            check_scores(eng, slot, lin, sums, w, h)
    eng.close()


@pytest.mark.parametrize("w,h,batch", [(333, 203, 3), (70, 38, 2), (1, 1, 1), (31, 96, 2), (257, 131, 1), (1920, 1080, 3), (1920, 1080, 9), (3840, 2160, 2)])
def test_fused_edge_kernel_is_bit_identical_with_the_two_passes(w, h, batch):
    """k_blur_edge_fused (both recurrences, the edge maps and their sums of the edge-only jobs in one kernel, beside the two
    passes on a second stream; what larger launches run by default) against the two-pass kernels: the same 108 sums, bit for
    bit -- one wave and four waves per workgroup would be chosen by the engine, forced here on every size --, repeated
    launches (the hand-off words of the previous launch are in place), and against the oracle on the small sizes"""
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=batch)
    frames = [nv12_frames(w, h, n) for n in range(batch)]
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    got = {}
    for name, variant in (("two_pass", F.TM_VARIANT_TWO_PASS_EDGE), ("fused", F.TM_VARIANT_FUSED_EDGE), ("fused_again", F.TM_VARIANT_FUSED_EDGE),
                          ("fused_split_rows", F.TM_VARIANT_FUSED_EDGE | F.TM_VARIANT_SPLIT_ROWS), ("default", F.TM_VARIANT_DEFAULT)):
        eng.set_variant(variant)
        assert eng.uses_fused_edge() == (name.startswith("fused") or (name == "default" and 2 * ((h + 31) // 32) * batch >= 400))
        eng.compute_async()
        eng.sync()
        got[name] = [eng.raw_sums(i).copy() for i in range(batch)]
    for i in range(batch):
        for name in got:
            assert np.array_equal(got[name][i], got["two_pass"][i]), (name, i)
    if w * h <= 333 * 203:
        eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
        eng.compute_async()
        eng.sync()
        for slot, (fr, fd) in enumerate(frames):
            lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
            sums, _ = O.ssimulacra2_sums(lin[0], lin[1], want_xyb=True)
            mask = (O.weights().reshape(3, 6, 6) != 0.0).transpose(1, 2, 0)
            np.testing.assert_allclose(eng.raw_sums(slot)[mask], np.asarray(sums).reshape(6, 6, 3)[mask], rtol=1e-12, atol=1e-300)
            assert abs(eng.scores(slot).ssimulacra2 - O.score_from_sums(sums, w, h)) <= 1e-9
    eng.debug_set_edge_epoch(0xFFFFFE)  # the tags hold 24 bits of the launch epoch: across the wrap (..FE, ..FF, 1, 2) nothing changes
    for _ in range(4):
        eng.compute_async()
        eng.sync()
        for i in range(batch):
            assert np.array_equal(eng.raw_sums(i), got["two_pass"][i])
    eng.set_graph(True)  # the two-stream sequence captured into a hipGraph and replayed (the epoch of the hand-off tags lives in device memory)
    eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    for _ in range(3):
        eng.compute_async()
        eng.sync()
        for i in range(batch):
            assert np.array_equal(eng.raw_sums(i), got["two_pass"][i])
    eng.set_graph(False)
    eng.set_full_sums(True)  # no edge-only job left: nothing for the fused kernel
    assert not eng.uses_fused_edge()
    eng.close()


@pytest.mark.parametrize("w,h", [(4100, 37), (37, 2100), (7680, 4320)])
def test_fused_edge_kernel_on_extreme_shapes(w, h):
    """wide and short (129 tiles, two bands), narrow and tall (two tiles, 66 bands = 17 groups chained through memory), 8K
    (240 tiles x 135 bands): the fused kernel's sums against the two-pass kernels', bit for bit"""
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    fr, fd = nv12_frames(w, h, 5)
    eng.set_pair(0, fr, fd)
    got = {}
    for name, variant in (("two_pass", F.TM_VARIANT_TWO_PASS_EDGE), ("fused", F.TM_VARIANT_FUSED_EDGE), ("fused_again", F.TM_VARIANT_FUSED_EDGE)):
        eng.set_variant(variant)
        eng.compute_async()
        eng.sync()
        got[name] = eng.raw_sums(0).copy()
    assert np.array_equal(got["fused"], got["two_pass"]) and np.array_equal(got["fused_again"], got["two_pass"])
    eng.close()


def test_two_engines_on_one_device_share_the_side_stream():
    """the fused kernel of the edge-only jobs runs on ONE side stream per device, shared by the engines: two engines driven from two
    host threads at the same time (the CLI's ping-pong pair, `--devices N` on a shared device) each get the sums they get alone"""
    import threading
    w, h, batch = 640, 360, 3
    engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=batch) for _ in range(2)]
    frames = [[nv12_frames(w, h, 10 * k + n) for n in range(batch)] for k in range(2)]
    want = []
    for k, eng in enumerate(engs):
        for slot, (fr, fd) in enumerate(frames[k]):
            eng.set_pair(slot, fr, fd)
        eng.set_variant(F.TM_VARIANT_TWO_PASS_EDGE)
        eng.compute_async(); eng.sync()
        want.append([eng.raw_sums(i).copy() for i in range(batch)])
        eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    errors = []

    def worker(k):
        try:
            for _ in range(40):
                engs[k].compute_async(); engs[k].sync()
                for i in range(batch):
                    if not np.array_equal(engs[k].raw_sums(i), want[k][i]):
                        raise AssertionError(f"engine {k} slot {i}")
        except Exception as exc:  # noqa: BLE001 -- reported below, in the main thread
            errors.append(exc)

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for eng in engs:
        eng.close()
    assert not errors, errors


def test_every_input_kind_matches_oracle():
    w, h = 94, 58
    rng = np.random.default_rng(5)
    r8, d8 = tm.synth.rgb8_pair(w, h)
    r16 = r8.astype(np.uint16) * 257
    d16 = (d8.astype(np.uint16) * 257 + rng.integers(0, 200, d8.shape)).astype(np.uint16)
    rf = rng.random((h, w, 3), dtype=np.float32)
    df = np.clip(rf + rng.normal(0, 0.03, rf.shape).astype(np.float32), 0, 1)
    frames = [p016_frames(w, h, 3), (tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8)), (tm.HwFrame.rgb(r16), tm.HwFrame.rgb(d16)),
              (tm.HwFrame.rgb(rf), tm.HwFrame.rgb(df)), (tm.HwFrame.linear(rf), tm.HwFrame.linear(df))]
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=len(frames), full_sums=True)
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    eng.compute_async()
    eng.sync()
    for slot, (fr, fd) in enumerate(frames):
        lin, sums = check_planes(eng, slot, fr, fd, w, h)
        check_scores(eng, slot, lin, sums, w, h)
    eng.close()


def test_device_resident_frames_and_slot_independence():
    torch = pytest.importorskip("torch")
    w, h = 200, 120
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=4)
    solo = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    frames = [nv12_frames(w, h, n) for n in range(4)]
    keep = []
    for slot, (fr, fd) in enumerate(frames):
        tr, td = torch.from_numpy(np.asarray(fr.data)).cuda(), torch.from_numpy(np.asarray(fd.data)).cuda()
        keep += [tr, td]
        eng.set_pair(slot, tm.HwFrame.nv12(tr, fr.pitch, fr.coded_height), tm.HwFrame.nv12(td, fd.pitch, fd.coded_height))
    torch.cuda.synchronize()
    eng.compute_async()
    eng.sync()
    for slot, (fr, fd) in enumerate(frames):
        one = solo.compute_one(fr, fd)
        assert np.array_equal(solo.raw_sums(0), eng.raw_sums(slot))  # bitwise: a slot never sees its neighbours
        assert one.ssimulacra2 == eng.scores(slot).ssimulacra2
    # determinism: a second run reproduces every bit
    first = [eng.raw_sums(i).copy() for i in range(4)]
    eng.compute_async()
    eng.sync()
    assert all(np.array_equal(first[i], eng.raw_sums(i)) for i in range(4))
    eng.close(); solo.close()


def ssim_sums_used(ssim, msssim):
    """(3 channels, 5 scales, [sum of l * cs, sum of cs]) bool: the sums a requested score reads -- SSIM the scale-0 mean of
    l * cs, MS-SSIM the mean of cs on scales 0..3 and of l * cs on scale 4 (Wang et al. 2003)"""
    m = np.zeros((3, 5, 2), bool)
    if ssim:
        m[:, 0, 0] = True
    if msssim:
        m[:, :4, 1] = True
        m[:, 4, 0] = True
    return m


def weight_mask():
    """(6 scales, 6 kinds, 3 channels) bool: sums with a non-zero weight in the reference's table [channel][scale][kind]"""
    return (O.weights().reshape(3, 6, 6) != 0.0).transpose(1, 2, 0)


@pytest.mark.parametrize("kind", ["nv12", "p016"])
def test_unaligned_device_surfaces_and_16bit_content(kind):
    """device-resident surfaces whose base pointer and pitch are not multiples of a sample pair (the ingest kernel's pair loads
    must fall back to single loads), and P016 words that use all 16 bits"""
    torch = pytest.importorskip("torch")
    w, h = 150, 70
    bps = 1 if kind == "nv12" else 2
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=3, full_sums=True)
    host, keep = [], []
    for slot in range(3):
        (rs, rp, rch), (ds, dp, dch) = (tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair)(w, h, slot)
        pair_h, pair_d = [], []
        for buf, pitch, ch in ((rs, rp, rch), (ds, dp, dch)):
            rows = ch * 3 // 2
            a = np.asarray(buf, np.uint8)[: pitch * rows].reshape(rows, pitch)
            newp = w * bps + (bps if slot else 2 * bps)  # slot 0: pair-aligned pitch, odd base; slots 1, 2: pitch = odd number of samples
            b = np.zeros((rows, newp), np.uint8)
            b[:, : w * bps] = a[:, : w * bps]
            if kind == "p016" and slot == 2:
                b.reshape(-1).view(np.uint16)[::3] |= 0x3F  # not 10-bit-in-16 any more
            flat = b.reshape(-1)
            off = bps if slot != 1 else 2 * bps  # slots 0, 2: base not pair-aligned
            t = torch.zeros(flat.size + off, dtype=torch.uint8, device="cuda")
            t[off:] = torch.from_numpy(flat).cuda()
            keep.append(t)
            mk = tm.HwFrame.nv12 if kind == "nv12" else tm.HwFrame.p016
            pair_h.append(mk(flat.copy(), newp, ch)); pair_d.append(mk(t[off:], newp, ch))
        host.append(pair_h)
        eng.set_pair(slot, pair_d[0], pair_d[1])
    eng.compute_async()
    eng.sync()
    for slot, (fr, fd) in enumerate(host):
        lin, sums = check_planes(eng, slot, fr, fd, w, h)
        check_scores(eng, slot, lin, sums, w, h)
    eng.close()


@pytest.mark.parametrize("w,h", [(70, 38), (333, 203), (1, 1), (129, 20), (640, 360)])
def test_weight_pruned_sums_equal_full_sums(w, h):
    """Default mode skips the 56 sums whose weight is 0.0: the 52 weighted sums and the score must be bit-identical
    to the full computation, and the full computation must match the oracle."""
    frames = [nv12_frames(w, h, n) for n in range(3)]
    res = []
    for full in (False, True):
        eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=3, full_sums=full)
        for slot, (fr, fd) in enumerate(frames):
            eng.set_pair(slot, fr, fd)
        eng.compute_async()
        eng.sync()
        res.append([(eng.raw_sums(i).copy(), eng.scores(i).ssimulacra2) for i in range(3)])
        eng.close()
    m = weight_mask()
    assert m.sum() == 52
    for slot, (fr, fd) in enumerate(frames):
        (sp, scp), (sf, scf) = res[0][slot], res[1][slot]
        assert np.array_equal(sp[m], sf[m])
        assert scp == scf
        want, sums = O.ssimulacra2_from_linear(oracle_linear(fr, w, h), oracle_linear(fd, w, h))
        np.testing.assert_allclose(sf, sums, rtol=1e-12, atol=1e-300)
        assert abs(scp - want) <= 1e-9


def test_identical_frames_property_full_hd():
    # size-independent property at BASELINE's full size: identical inputs -> SSIM map exactly 0 at every
    # scale/channel, PSNR = +inf, score just below 100 (edge-term rounding residue, see oracle pins)
    w, h = 1920, 1080
    fr, _ = nv12_frames(w, h, 5)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=1, full_sums=True)
    s = eng.compute_one(fr, fr)
    sums = eng.raw_sums(0)
    assert np.all(sums[:, 0, :] == 0.0) and np.all(sums[:, 3, :] == 0.0)
    assert 99.9 < s.ssimulacra2 <= 100.0
    assert eng.sse(0) == 0 and s.psnr == float("inf")
    eng.close()


def test_full_hd_pair_against_oracle():
    w, h = 1920, 1080
    fr, fd = nv12_frames(w, h, 2)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
    eng.set_pair(0, fr, fd)
    eng.set_pair(1, fd, fr)  # swapped roles in the neighbouring slot
    eng.compute_async()
    eng.sync()
    lin, sums = check_planes(eng, 0, fr, fd, w, h, scales=[0, 5])
    check_scores(eng, 0, lin, sums, w, h)
    assert eng.sse(1) == eng.sse(0)  # SSE is symmetric in its arguments
    eng.close()


def test_4k_p016_pair_against_oracle():
    w, h = 3840, 2160
    fr, fd = p016_frames(w, h, 1)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)  # default: only the sums that carry weight
    got = eng.compute_one(fr, fd)
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    want, sums = O.ssimulacra2_from_linear(lin[0], lin[1])
    m = weight_mask()
    np.testing.assert_allclose(eng.raw_sums(0)[m], sums[m], rtol=1e-12, atol=1e-300)
    assert abs(got.ssimulacra2 - want) <= 1e-9
    eng.set_full_sums(True)
    got_full = eng.compute_one(fr, fd)
    np.testing.assert_allclose(eng.raw_sums(0), sums, rtol=1e-12, atol=1e-300)
    assert got_full.ssimulacra2 == got.ssimulacra2
    eng.close()


def test_4k_p016_fused_psnr_msssim_ssimulacra2_against_oracle():
    """BASELINE.json configs[4] on one GPU: 4K P016, `-m psnr -m msssim -m ssimulacra2` in one fused pass, batch > 1 (the second
    slot holds the swapped pair).  SSIMULACRA2 sums 1e-12 / score 1e-9, PSNR bit-exact, MS-SSIM sums 1e-12 / score 1e-6."""
    w, h = 3840, 2160
    fr, fd = p016_frames(w, h, 2)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, msssim=True), batch=2)
    eng.set_pair(0, fr, fd)
    eng.set_pair(1, fd, fr)
    eng.compute_async()
    eng.sync()
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    m = weight_mask()
    for slot, (a, b) in enumerate([(lin[0], lin[1]), (lin[1], lin[0])]):
        want, sums = O.ssimulacra2_from_linear(a, b)
        got = eng.scores(slot)
        np.testing.assert_allclose(eng.raw_sums(slot)[m], sums[m], rtol=1e-12, atol=1e-300)
        assert abs(got.ssimulacra2 - want) <= 1e-9
        sse, psnr = O.psnr(a, b)
        assert eng.sse(slot) == sse and got.psnr == psnr
        _, want_ms, ssums = O.ssim_msssim(a, b)
        used = ssim_sums_used(ssim=False, msssim=True)
        np.testing.assert_allclose(eng.ssim_sums(slot)[used], ssums[used], rtol=1e-12, atol=1e-300)
        assert abs(got.msssim - want_ms) <= 1e-6 and 0.0 < got.msssim <= 1.0
        assert got.ssim is None
    assert eng.scores(0).ssimulacra2 != eng.scores(1).ssimulacra2
    eng.close()


@pytest.mark.parametrize("bits,w,h", [(8, 333, 203), (10, 333, 203), (8, 70, 38), (10, 1920, 1080), (12, 129, 20)])
def test_planar_i420_is_bit_identical_with_the_repacked_biplanar_surface(bits, w, h):
    """tm_engine_set_frame_i420 (planar 4:2:0 as files deliver it: u8, or the value in the low bits of little-endian u16) against
    tm_engine_set_frame_{nv12,p016} on the surface the same samples are repacked into -- the reference's only YUV contract --
    and against the oracle: same raw sums, SSE and scores, from host memory and from device memory."""
    import torch
    dt = np.uint8 if bits == 8 else np.uint16
    frames = []
    for n in range(2):
        ref, dis = tm.synth.yuv420_pair(w, h, n + 1, 8 if bits == 8 else 10)
        if bits == 12:
            ref, dis = tuple(p * 4 + 1 for p in ref), tuple(p * 4 + 2 for p in dis)
        frames.append((ref, dis))
    planar = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
    packed = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
    keep = []
    for slot, (ref, dis) in enumerate(frames):
        for side, planes in enumerate((ref, dis)):
            pl = [np.ascontiguousarray(p.astype(dt)) for p in planes]
            if slot == 1:  # device-resident planes (zero copy), with a pitch wider than the row
                pl = [torch.from_numpy(np.pad(p, ((0, 0), (0, 6)))).cuda()[:, :p.shape[1]] for p in pl]
                keep.append(pl)
            planar.set_frame(slot, side, tm.HwFrame.i420(pl[0], pl[1], pl[2], bits=bits, matrix=tm.ColorMatrix(slot)))
            surf, pitch, ch = tm.synth.pack_biplanar(planes, w, h, bits)
            mk = tm.HwFrame.nv12 if bits == 8 else tm.HwFrame.p016
            packed.set_frame(slot, side, mk(surf, pitch, ch, tm.ColorMatrix(slot)))
    for e in (planar, packed):
        e.compute_async(); e.sync()
    for slot, (ref, dis) in enumerate(frames):
        assert np.array_equal(planar.raw_sums(slot), packed.raw_sums(slot))
        assert planar.sse(slot) == packed.sse(slot) and planar.scores(slot) == packed.scores(slot)
        lin = []
        for planes in (ref, dis):
            surf, pitch, ch = tm.synth.pack_biplanar(planes, w, h, bits)
            lin.append(O.yuv420_biplanar_to_linear(surf, pitch, ch, w, h, 8 if bits == 8 else 16, slot))
        if w * h <= 640 * 360:
            want, sums = O.ssimulacra2_from_linear(lin[0], lin[1])
            np.testing.assert_allclose(planar.raw_sums(slot), sums, rtol=1e-12, atol=1e-300)
            assert abs(planar.scores(slot).ssimulacra2 - want) <= 1e-9
        assert planar.sse(slot) == O.psnr(lin[0], lin[1])[0]
    with pytest.raises(tm.TmError) as ei:  # full range is todo!() in the reference for every YUV kind
        planar.set_frame(0, 0, tm.HwFrame.i420(*[np.ascontiguousarray(p.astype(dt)) for p in frames[0][0]], bits=bits, full_range=True))
    assert ei.value.code == F.TM_ERR_UNSUPPORTED
    planar.close(); packed.close()


@pytest.mark.parametrize("w,h", [(333, 203), (70, 38), (1, 1), (129, 20), (770, 67), (1920, 1080), (3840, 2160)])
def test_packed_10_bit_upload_kind_is_bit_identical_with_planar_10_bit(w, h):
    """tm_engine_set_frame_i420p10 -- three 10-bit samples per 32-bit word, 10.7 instead of 16 bits per sample over PCIe --
    against tm_engine_set_frame_i420(bits = 10) on the same samples: raw sums, SSE and scores bit for bit (the row-walking kernel, the tile
    kernel, the reference pipeline; mixed with another kind in one launch), from pageable, page-locked and device memory, as separate
    planes with a wider pitch and as one tight picture (one linear copy), packed by numpy (the header's sentence) and by tm_p10_pack_rows;
    small sizes against the oracle."""
    import torch
    L = F.lib()
    cw, ch = (w + 1) // 2, (h + 1) // 2
    B = 2 if w * h > 2000 * 1000 else 3
    pairs = [tm.synth.yuv420_pair(w, h, n + 7, 10) for n in range(B)]
    m = tm.Metrics(ssimulacra2=True, psnr=True)
    planar = tm.TurboMetrics(w, h, m, batch=B, full_sums=True)
    for slot, (ref, dis) in enumerate(pairs):
        for side, planes in enumerate((ref, dis)):
            planar.set_frame(slot, side, tm.HwFrame.i420(*(np.ascontiguousarray(p.astype(np.uint16)) for p in planes), bits=10, matrix=tm.ColorMatrix(slot % 3)))
    planar.compute_async(); planar.sync()
    want = [(planar.raw_sums(s).copy(), planar.sse(s), planar.scores(s)) for s in range(B)]

    def lib_pack(plane, pitch_words):  # the library's packer, from u16 rows with junk above bit 9
        src = np.ascontiguousarray((plane.astype(np.int64) | 0xFC00).astype(np.uint16))
        dst = np.zeros((plane.shape[0], pitch_words), np.uint32)
        L.tm_p10_pack_rows(src.ctypes.data, src.strides[0], plane.shape[1], plane.shape[0], dst.ctypes.data, dst.strides[0])
        return dst

    def tight(planes, pinned):  # Y, Cb, Cr words back to back without row padding: goes up as one linear copy
        flat = np.concatenate([tm.synth.p10_pack_plane(p).reshape(-1) for p in planes])
        t = torch.from_numpy(flat.view(np.int32))
        t = t.pin_memory() if pinned else t
        wy, wc = tm.synth.p10_row_words(w), tm.synth.p10_row_words(cw)
        return t[:h * wy].view(h, wy), t[h * wy:h * wy + ch * wc].view(ch, wc), t[h * wy + ch * wc:].view(ch, wc)

    makers = {
        "numpy_pageable": lambda pl, slot: [tm.synth.p10_pack_plane(p, tm.synth.p10_row_words(p.shape[1]) + 2 * slot) for p in pl],
        "library_packer": lambda pl, slot: [lib_pack(p, tm.synth.p10_row_words(p.shape[1]) + 2) for p in pl],
        "device": lambda pl, slot: [torch.from_numpy(tm.synth.p10_pack_plane(p, tm.synth.p10_row_words(p.shape[1]) + 4).view(np.int32)).cuda() for p in pl],
        "tight_pinned": lambda pl, slot: tight(pl, True),
        "tight_pageable": lambda pl, slot: tight(pl, False),
    }
    for name, make in makers.items():
        for variant in ((0, F.TM_VARIANT_TILE_INGEST, F.TM_VARIANT_REFERENCE) if name == "numpy_pageable" and w * h <= 1920 * 1080 else (0,)):
            eng = tm.TurboMetrics(w, h, m, batch=B, full_sums=True)
            eng.set_variant(variant)
            eng.debug_set_param(F.TM_DBG_LINEAR_UPLOAD, 1)
            keep = []
            for slot, (ref, dis) in enumerate(pairs):
                for side, planes in enumerate((ref, dis)):
                    pl = make(planes, slot); keep.append(pl)
                    eng.set_frame(slot, side, tm.HwFrame.i420p10(pl[0], pl[1], pl[2], matrix=tm.ColorMatrix(slot % 3)))
            eng.compute_async(); eng.sync()
            for s in range(B):
                assert np.array_equal(eng.raw_sums(s), want[s][0]) and eng.sse(s) == want[s][1] and eng.scores(s) == want[s][2], (name, variant, s)
            if name == "numpy_pageable" and variant == 0 and w * h <= 640 * 360:
                fr = tm.HwFrame.p016(*tm.synth.pack_biplanar(pairs[0][0], w, h, 10)); fd = tm.HwFrame.p016(*tm.synth.pack_biplanar(pairs[0][1], w, h, 10))
                lin, sums = check_planes(eng, 0, fr, fd, w, h)
                check_scores(eng, 0, lin, sums, w, h)
            if name == "numpy_pageable" and variant == 0:  # one launch of two kinds: per-frame dispatch
                pl = [np.ascontiguousarray(p.astype(np.uint16)) for p in pairs[1][0]]
                eng.set_frame(1, 0, tm.HwFrame.i420(pl[0], pl[1], pl[2], bits=10, matrix=tm.ColorMatrix(1)))
                eng.compute_async(); eng.sync()
                for s in range(B):
                    assert np.array_equal(eng.raw_sums(s), want[s][0]) and eng.sse(s) == want[s][1], ("mixed", s)
            eng.close()
    e1 = tm.TurboMetrics(w, h, m, batch=1)
    pk = [tm.synth.p10_pack_plane(p) for p in pairs[0][0]]
    if h > 1:  # (a one-row array has no pitch to speak of)
        with pytest.raises(tm.TmError) as ei:  # a pitch below one packed row
            e1.set_frame(0, 0, tm.HwFrame.i420p10(np.ascontiguousarray(pk[0][:, :-2]), pk[1], pk[2]))
        assert ei.value.code == F.TM_ERR_INVALID_ARG
    with pytest.raises(tm.TmError) as ei:
        e1.set_frame(0, 0, tm.HwFrame.i420p10(pk[0], pk[1], pk[2], full_range=True))
    assert ei.value.code == F.TM_ERR_UNSUPPORTED
    e1.close(); planar.close()


@pytest.mark.parametrize("form", ["nv12_surfaces", "i420_tight", "p10_tight"])
def test_back_to_back_page_locked_frames_share_a_dma_and_nothing_changes(form):
    """round 6 (second half): page-locked frames that lie back to back in the caller's memory -- a decoder's surface pool, the
    CLI's ring -- land back to back in the engine's staging arena, and two of them go up as ONE DMA (a copy is held back until the next one
    of its stream is known; fences, launches and syncs flush).  Same sums as device-resident frames whatever the hand-over order: in slot order
    (pairs merge), in reverse (nothing merges), a slot handed over twice, a pageable frame behind a held page-locked one, fences in between,
    merging off (TM_DBG_UPLOAD_MERGE 0), one upload stream."""
    import torch
    w, h, B = 640, 360, 6
    cw, ch = (w + 1) // 2, (h + 1) // 2
    m = tm.Metrics(ssimulacra2=True, psnr=True)
    pairs = [tm.synth.yuv420_pair(w, h, n + 11, 10 if form == "p10_tight" else 8) for n in range(B + 1)]
    base = tm.TurboMetrics(w, h, m, batch=B, full_sums=True)

    def dev_frame(planes):
        if form == "p10_tight":
            return tm.HwFrame.i420(*(torch.from_numpy(np.ascontiguousarray(p.astype(np.uint16))).cuda() for p in planes), bits=10)
        return tm.HwFrame.i420(*(torch.from_numpy(np.ascontiguousarray(p.astype(np.uint8))).cuda() for p in planes), bits=8)
    keep = []
    for slot in range(B):
        fr = [dev_frame(pairs[slot][side]) for side in range(2)]; keep.append(fr)
        base.set_pair(slot, fr[0], fr[1])
    base.compute_async(); base.sync()
    want = [(base.raw_sums(s).copy(), base.sse(s)) for s in range(B)]
    other = tm.TurboMetrics(w, h, m, batch=1, full_sums=True)
    fr7 = [dev_frame(pairs[B][side]) for side in range(2)]
    other.compute_one(fr7[0], fr7[1])
    want7 = (other.raw_sums(0).copy(), other.sse(0))

    # one page-locked pool per side, frames back to back
    if form == "nv12_surfaces":
        surf = [[tm.synth.pack_biplanar(pairs[n][side], w, h, 8, coded_height=h) for n in range(B + 1)] for side in range(2)]
        nbytes = len(surf[0][0][0])
        mk = lambda side, n, t: tm.HwFrame.nv12(t, surf[side][n][1], surf[side][n][2])
        blob = lambda side, n: surf[side][n][0]
    elif form == "i420_tight":
        blob = lambda side, n: np.concatenate([p.astype(np.uint8).ravel() for p in pairs[n][side]])
        nbytes = w * h + 2 * cw * ch
        mk = lambda side, n, t: tm.HwFrame.i420(t[:w * h].view(h, w), t[w * h:w * h + cw * ch].view(ch, cw), t[w * h + cw * ch:].view(ch, cw), bits=8)
    else:
        wy, wc = tm.synth.p10_row_words(w), tm.synth.p10_row_words(cw)
        blob = lambda side, n: np.concatenate([tm.synth.p10_pack_plane(p).ravel() for p in pairs[n][side]]).view(np.uint8)
        nbytes = (h * wy + 2 * ch * wc) * 4
        def mk(side, n, t):
            t32 = t.view(torch.int32)
            return tm.HwFrame.i420p10(t32[:h * wy].view(h, wy), t32[h * wy:h * wy + ch * wc].view(ch, wc), t32[h * wy + ch * wc:].view(ch, wc))
    pools = [torch.empty(nbytes * (B + 1), dtype=torch.uint8).pin_memory() for _ in range(2)]
    for side in range(2):
        for n in range(B + 1):
            pools[side][n * nbytes:(n + 1) * nbytes].copy_(torch.from_numpy(np.ascontiguousarray(blob(side, n))))
    frame = lambda side, n: mk(side, n, pools[side][n * nbytes:(n + 1) * nbytes])

    def check(eng, expect):
        eng.compute_async(); eng.sync()
        for s in range(B):
            assert np.array_equal(eng.raw_sums(s), expect[s][0]) and eng.sse(s) == expect[s][1], (form, s)
    for merge, streams in ((14 << 20, 2), (0, 2), (7 << 20, 1), (1 << 30, 2)):
        eng = tm.TurboMetrics(w, h, m, batch=B, full_sums=True)
        eng.debug_set_param(F.TM_DBG_LINEAR_UPLOAD, 1)
        eng.debug_set_param(F.TM_DBG_UPLOAD_MERGE, merge)
        eng.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, streams)
        for slot in range(B):  # in slot order: neighbours share a DMA
            eng.set_pair(slot, frame(0, slot), frame(1, slot))
        check(eng, want)
        for slot in reversed(range(B)):  # in reverse: nothing to merge, every copy is held once and flushed by the next
            eng.set_pair(slot, frame(0, slot), frame(1, slot))
            if slot == 3:
                tok = eng.upload_fence()
                assert eng.upload_done(tok, block=True)
        check(eng, want)
        # slot 2 handed over twice (the wrong picture first), slot 4's distorted side replaced from pageable memory behind a held page-locked copy
        for slot in range(B):
            if slot == 2:
                eng.set_pair(2, frame(0, B), frame(1, B))
            eng.set_pair(slot, frame(0, slot), frame(1, slot))
            if slot == 4:
                eng.set_frame(4, F.TM_SIDE_DIS, frame(1, B))
                pageable = mk(1, 4, torch.from_numpy(np.ascontiguousarray(blob(1, 4)).copy()))
                eng.set_frame(4, F.TM_SIDE_DIS, pageable)
        check(eng, want)
        # every slot the seventh pair, fences after each hand-over
        toks = []
        for slot in range(B):
            eng.set_pair(slot, frame(0, B), frame(1, B))
            toks.append(eng.upload_fence())
        assert all(eng.upload_done(t, block=True) for t in toks)
        check(eng, [want7] * B)
        eng.close()
    base.close(); other.close()


def test_small_repeating_launches_replay_a_graph_by_themselves_with_the_same_bits():
    """round 6: launches of up to four pairs that come four times in a row replay a captured hipGraph by default (the reference's compute_one loop:
    +3-4 %); a change of shape, of a setting or of the frames' kind goes back to direct launches and captures again -- the sums never change.
    tm_engine_set_graph(e, 0) keeps every launch direct."""
    w, h = 416, 240
    m = tm.Metrics(ssimulacra2=True, psnr=True)
    eng = tm.TurboMetrics(w, h, m, batch=4, full_sums=True)
    ref = tm.TurboMetrics(w, h, m, batch=4, full_sums=True)
    ref.set_graph(False)
    eng.set_graph(True); eng.set_graph(None)  # (forced, then the default again)
    frames = [nv12_frames(w, h, 30 + i) for i in range(6)] + [p016_frames(w, h, 40 + i) for i in range(2)]
    rng = np.random.default_rng(5)
    n, picks = 1, [0]
    for step in range(120):
        if step % 9 == 0:  # a new shape every nine launches: the first four of each run go out directly
            n = int(rng.integers(1, 5))
        if step % 9 == 5:
            eng.set_full_sums(bool(step % 2)); ref.set_full_sums(bool(step % 2))
        kind16 = step % 27 >= 18
        picks = [int(rng.integers(6, 8)) if kind16 else int(rng.integers(0, 6)) for _ in range(n)]
        for e in (eng, ref):
            for slot, k in enumerate(picks):
                e.set_pair(slot, *frames[k])
            e.compute_async(n); e.sync()
        for slot in range(n):
            assert np.array_equal(eng.raw_sums(slot), ref.raw_sums(slot)) and eng.sse(slot) == ref.sse(slot), (step, n, slot)
    eng.close(); ref.close()


def test_engine_streams_get_hardware_queues_of_their_own_behind_other_peoples_streams(capfd):
    """round 6: the runtime binds a stream to one of its four hardware queues when the stream is created, and kernels of two streams on
    one queue run one after the other.  With an RCCL communicator (or any other owner of streams) created BEFORE the engine, the side
    stream used to land on the engine's own queue: the fused kernel ran behind the column pass and the headline lost 20 %
    (profiles/r06y9_dist_first.log).  The library now times where a new stream landed and asks for another one when that queue is taken:
    behind six foreign streams the engine, the side stream, the upload stream and a second engine still sit on four different queues."""
    torch = pytest.importorskip("torch")
    import re
    foreign = [torch.cuda.Stream() for _ in range(6)]
    for st in foreign:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
    F.lib().tm_set_debug_log(1)
    try:
        a = tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=8)
        b = tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=8)
    finally:
        F.lib().tm_set_debug_log(0)
    err = capfd.readouterr().err
    rows = re.findall(r"stream for (an engine|a small engine|the fused kernel|uploads) on hardware queue (-?\d+) of (\d+) \(cost (\d+)", err)
    assert [r[0] for r in rows].count("an engine") == 2, err
    assert all(int(r[2]) >= 4 for r in rows), err            # the four queues of the default runtime were found
    assert all(int(r[3]) == 0 for r in rows), err            # nobody shares a queue ...
    queues = [int(r[1]) for r in rows]
    assert len(set(queues)) == len(queues) and min(queues) >= 0, err  # ... (engine, [side, upload when this test made them], second engine)
    fr, fd = nv12_frames(1920, 1080, 1)
    for slot in range(8):
        a.set_pair(slot, fr, fd); b.set_pair(slot, fd, fr)
    a.compute_async(8); b.compute_async(8); a.sync(); b.sync()
    one = tm.TurboMetrics(1920, 1080, tm.Metrics(ssimulacra2=True), batch=1)
    assert a.scores(7) == one.compute_one(fr, fd) and b.scores(0) == one.compute_one(fd, fr)
    a.close(); b.close(); one.close()


def test_deferred_depth_three_to_eight_pairs_in_flight_with_compute_ones_scores():
    """round 6: set_deferred_depth(d) -- d one-pair launches in flight on d engines (each created when its turn first comes), pair k
    collected after pair k + d - 1 went in; the depth changes between runs with pairs still in flight (they are finished and stay
    collectable), engines beyond the new depth are freed.  FrameScores are compute_one's, bit for bit, at every depth."""
    w, h = 416, 240
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=True), batch=1)
    eng.set_full_sums(True)  # (a setting made before the further engines exist: they start with it)
    frames = [nv12_frames(w, h, 50 + i) for i in range(7)]
    want = [eng.compute_one(fr, fd) for fr, fd in frames]
    mem_one = eng.mem_usage()
    stale = []
    for depth in (3, 8, 2, 5, 4):
        eng.set_deferred_depth(depth, create_now=depth in (8, 5))
        got, tickets = [], []
        for k in range(3 * depth + 4):
            tickets.append((eng.compute_one_deferred(*frames[k % 7]), k % 7))
            if len(tickets) >= depth:
                t, i = tickets.pop(0)
                got.append((eng.collect(t), i))
        assert all(s == want[i] for s, i in got), depth
        assert len(tickets) == depth - 1
        stale.append(tickets)  # left in flight across the change of depth
    for tickets in stale:
        for t, i in reversed(tickets):
            assert eng.collect(t) == want[i]
    with pytest.raises(tm.TmError):
        eng.collect(stale[0][0][0])
    for bad in (0, 1, 9):
        with pytest.raises(tm.TmError):
            eng.set_deferred_depth(bad)
    eng.set_channel_mode(True)  # a setting reaches every engine of the turn
    first = eng.compute_one(*frames[0])
    assert first != want[0]
    ts = [eng.compute_one_deferred(*frames[0]) for _ in range(4)]
    assert [eng.collect(t) for t in ts] == [first] * 4
    assert eng.mem_usage() == mem_one  # (mem_usage is the engine's own; the further engines are the mirror's)
    eng.close()


def test_a_captured_launch_keeps_its_bits_when_engines_are_created_after_the_capture():
    """round 6: the HIP 7.0 runtime PyTorch bundles (the one this process runs on: conftest imports torch first) replayed the SSE reset --
    then a memset node -- with stale arguments once ANOTHER engine had been created after the capture: PSNR garbage, every other sum right
    (profiles/r06x_graph_memset.log).  The reset is a kernel now; replays before and after new engines come and go give the direct
    launch's bits, with the replay forced and with the default mode."""
    w, h = 640, 360
    m = tm.Metrics(ssimulacra2=True, psnr=True, ssim=True)
    frames = [nv12_frames(w, h, i) for i in range(3)]
    ref = tm.TurboMetrics(w, h, m, batch=1)
    ref.set_graph(False)
    want = [ref.compute_one(*f) for f in frames]
    for forced in (True, False):
        eng = tm.TurboMetrics(w, h, m, batch=1)
        if forced:
            eng.set_graph(True)
        assert [eng.compute_one(*frames[k % 3]) for k in range(6)] == want * 2  # captured by now in either mode
        others = []
        for k in range(6):
            others.append(tm.TurboMetrics(w + 64 * k, h, m, batch=1 + k % 2))
            assert eng.compute_one(*frames[k % 3]) == want[k % 3], (forced, k)
            if k % 2:
                others.pop(0).close()
                assert eng.compute_one(*frames[k % 3]) == want[k % 3], (forced, k)
        for o in others:
            o.close()
        eng.close()
    ref.close()


def test_8k_pair_against_oracle():
    """7680x4320: one slot's arenas pass 2 GB (several kernels carry 32-bit lane offsets inside a plane), two slots so that the
    second one starts beyond 4 GB of the pass-1 arena"""
    w, h = 7680, 4320
    fr, fd = nv12_frames(w, h, 3)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2)
    eng.set_pair(0, fd, fr)
    eng.set_pair(1, fr, fd)
    eng.compute_async()
    eng.sync()
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    want, sums = O.ssimulacra2_from_linear(lin[0], lin[1])
    m = weight_mask()
    np.testing.assert_allclose(eng.raw_sums(1)[m], sums[m], rtol=1e-12, atol=1e-300)
    assert abs(eng.scores(1).ssimulacra2 - want) <= 1e-9
    assert eng.scores(0).ssimulacra2 != eng.scores(1).ssimulacra2  # the metric is not symmetric: slot 0 really is the swapped pair
    assert eng.sse(0) == eng.sse(1)
    eng.close()


def test_reference_mirror_types_and_errors():
    w, h = 96, 64
    r8, d8 = tm.synth.rgb8_pair(w, h)
    ss = tm.Ssimulacra2(w, h)
    got = ss.compute_srgb_sync(r8, d8)
    want, _ = O.ssimulacra2_from_linear(O.rgb8_to_linear(r8), O.rgb8_to_linear(d8))
    assert abs(got - want) <= 1e-9
    assert ss.mem_usage() > 270 * w * h / 4
    ss.close()
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=2)
    fr, fd = nv12_frames(w, h, 0)
    # reference todo!() territory -> TM_ERR_UNSUPPORTED (cuda-colorspace/src/lib.rs:45-52)
    bad = tm.HwFrame.nv12(fr.data, fr.pitch, fr.coded_height, full_range=True)
    with pytest.raises(tm.TmError) as ei:
        eng.set_frame(0, 0, bad)
    assert ei.value.code == F.TM_ERR_UNSUPPORTED
    with pytest.raises(tm.TmError) as ei:  # a pitch the 32-bit lane offsets of the ingest kernel cannot address (checked before any byte is read)
        eng.set_frame(0, 0, tm.HwFrame.nv12(fr.data, 1 << 24, fr.coded_height))
    assert ei.value.code == F.TM_ERR_INVALID_ARG
    with pytest.raises(tm.TmError) as ei:  # compute before both sides of every slot are set
        eng.compute_async(2)
    assert ei.value.code == F.TM_ERR_STATE
    with pytest.raises(tm.TmError):
        eng.scores(0)
    # compute_all == reference frame selection, batched
    frames = [nv12_frames(w, h, n) for n in range(5)]
    res = eng.compute_all([f[0] for f in frames], [f[1] for f in frames], tm.engine.Options(every=2))
    assert len(res) == 3
    solo = [eng.compute_one(*frames[i]).ssimulacra2 for i in (0, 2, 4)]
    assert [r.ssimulacra2 for r in res] == solo
    eng.close()


@pytest.mark.parametrize("w,h,metrics", [(200, 180, "both"), (64, 40, "ssim"), (333, 203, "both"), (1920, 1080, "both")])
def test_ssim_and_msssim_match_oracle(w, h, metrics):
    """SSIM / MS-SSIM are BUILD-DEFINED (NPP's are closed and unpinned): HIP vs the oracle's statement of the same
    definition.  Per-scale sums to 1e-12, scores to 1e-6 (they are single floats by contract), fused with the other metrics."""
    m = tm.Metrics(ssimulacra2=True, psnr=True, ssim=True, msssim=(metrics == "both"))
    n = 2 if w * h > 10 ** 6 else 3
    eng = tm.TurboMetrics(w, h, m, batch=n)
    only = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=n)
    frames = [nv12_frames(w, h, i) for i in range(n)]
    if n == 3:
        r8, d8 = tm.synth.rgb8_pair(w, h)
        frames[2] = (tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8))
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
        only.set_pair(slot, fr, fd)
    eng.compute_async(); eng.sync()
    only.compute_async(); only.sync()
    used = ssim_sums_used(ssim=True, msssim=(metrics == "both"))
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        want_ssim, want_ms, sums = O.ssim_msssim(lin[0], lin[1])
        got = eng.scores(slot)
        gs = eng.ssim_sums(slot)
        np.testing.assert_allclose(gs[used], sums[used], rtol=1e-12, atol=1e-300)
        assert np.all(gs[..., 0][~used[..., 0]] == 0.0)  # the luminance term is only evaluated where a score reads it
        assert abs(got.ssim - want_ssim) <= 1e-6 and 0.0 < got.ssim <= 1.0
        if metrics == "both":
            assert abs(got.msssim - want_ms) <= 1e-6 and 0.0 < got.msssim <= 1.0
        else:
            assert got.msssim is None
        # the fused pass leaves the other metrics untouched
        assert got.ssimulacra2 == only.scores(slot).ssimulacra2
        assert got.psnr == O.psnr(lin[0], lin[1])[1]
    # tm_engine_set_full_sums(1): every [channel][scale][sum of l * cs, sum of cs] entry of the scales that were run
    eng.set_full_sums(True)
    eng.compute_async(); eng.sync()
    nsc = 5 if metrics == "both" else 1
    for slot, (fr, fd) in enumerate(frames):
        _, _, sums = O.ssim_msssim(oracle_linear(fr, w, h), oracle_linear(fd, w, h))
        np.testing.assert_allclose(eng.ssim_sums(slot)[:, :nsc], sums[:, :nsc], rtol=1e-12, atol=1e-300)
    eng.close(); only.close()


def test_ssim_and_msssim_match_the_float64_twin():
    """The HIP kernels against the SECOND statement of the two metrics (oracle/twin_ssim.py: float64 scipy written from the
    papers, independent of oracle/tm_ssim.c): per-scale window means 5e-6 relative, scores 1e-6 (tests/test_ssim_twin.py
    explains the tolerances)."""
    from oracle import twin_ssim as T
    for w, h in [(333, 203), (640, 360)]:
        eng = tm.TurboMetrics(w, h, tm.Metrics(ssim=True, msssim=True), batch=1)
        fr, fd = nv12_frames(w, h, 4)
        got = eng.compute_one(fr, fd)
        eng.set_full_sums(True)
        eng.compute_async(); eng.sync()
        lr, ld = oracle_linear(fr, w, h), oracle_linear(fd, w, h)
        want = T.scale_means(lr, ld, 5, odd="drop")
        counts, sw, sh = [], w, h
        for _ in range(5):
            counts.append((sw - 10) * (sh - 10)); sw //= 2; sh //= 2
        np.testing.assert_allclose(eng.ssim_sums(0) / np.asarray(counts, np.float64)[None, :, None], want, rtol=5e-6)
        assert abs(got.ssim - T.ssim(lr, ld)) <= 1e-6 and abs(got.msssim - T.msssim(lr, ld)) <= 1e-6
        eng.close()


def test_ssim_of_identical_frames_is_one_and_size_limits():
    w, h = 192, 176
    fr, _ = nv12_frames(w, h, 1)
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssim=True, msssim=True), batch=1)
    s = eng.compute_one(fr, fr)
    assert s.ssim == 1.0 and s.msssim == 1.0 and s.ssimulacra2 is None and s.psnr is None
    eng.close()
    with pytest.raises(tm.TmError) as ei:      # fifth dyadic scale smaller than the 11x11 window
        tm.TurboMetrics(175, 300, tm.Metrics(msssim=True), batch=1)
    assert ei.value.code == F.TM_ERR_UNSUPPORTED
    with pytest.raises(tm.TmError):
        tm.TurboMetrics(10, 300, tm.Metrics(ssim=True), batch=1)


def test_per_channel_values_and_first_channel_mode():
    """NPP's C3 quality functions may report one value per channel, of which the reference's 4-byte read-back would keep
    the first: the per-channel SSE / SSIM / MS-SSIM are exposed, and TM_CHANNELS_FIRST reports
    channel 0 instead of the pooled / averaged value.  Checked against numpy on the oracle's quantised frames."""
    w, h = 208, 192
    fr, fd = nv12_frames(w, h, 3)
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    qa, qb = O.quantize_u8(lin[0]).astype(np.int64), O.quantize_u8(lin[1]).astype(np.int64)
    sse_c = [int(((qa[c] - qb[c]) ** 2).sum()) for c in range(3)]
    eng = tm.TurboMetrics(w, h, tm.Metrics(psnr=True, ssim=True, msssim=True), batch=1)
    pooled = eng.compute_one(fr, fd)
    assert eng.sse_channels(0) == sse_c and eng.sse(0) == sum(sse_c)
    sums = eng.ssim_sums(0)
    L = tm.ffi.lib()
    import ctypes as C
    sp = np.ascontiguousarray(sums.ravel()).ctypes.data_as(C.POINTER(C.c_double))
    ssim_c = [L.tm_ssim_channel_from_sums(sp, w, h, c) for c in range(3)]
    ms_c = [L.tm_msssim_channel_from_sums(sp, w, h, c) for c in range(3)]
    assert abs(pooled.ssim - np.mean(ssim_c)) < 1e-6 and abs(pooled.msssim - np.mean(ms_c)) < 1e-6
    eng.set_channel_mode(True)
    first = eng.scores(0)
    assert first.psnr == L.tm_psnr_from_sse(sse_c[0], w * h) and first.psnr == float(np.float32(10 * np.log10(255.0 ** 2 * w * h / sse_c[0])))
    assert first.ssim == ssim_c[0] and first.msssim == ms_c[0]
    assert pooled.psnr == L.tm_psnr_from_sse(sum(sse_c), 3 * w * h) == O.psnr(lin[0], lin[1])[1]
    eng.close()


def test_placement_search_changes_nothing_but_the_address():
    """tm_set_placement_candidates: engine creation keeps the fastest of a few allocations of the pass-1 arena (only for arenas
    of 1 GiB and more: 8 slots of 1080p).  Scores, raw sums and the memory accounting must not depend on it."""
    w, h, B = 1920, 1080, 8
    frames = [nv12_frames(w, h, n) for n in range(2)]
    out = []
    try:
        for cand in (1, 3):
            tm.set_placement_candidates(cand)
            eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
            for slot in range(B):
                eng.set_pair(slot, *frames[slot % 2])
            eng.compute_async()
            eng.sync()
            out.append((eng.mem_usage(), [eng.scores(s).ssimulacra2 for s in range(B)], [eng.raw_sums(s).tobytes() for s in range(B)]))
            eng.close()
    finally:
        tm.set_placement_candidates(8)
    assert out[0] == out[1]


def test_graph_replay_equals_direct_launches():
    """The per-batch sequence is replayed from a captured hipGraph; frame pointers, batch size, input kind and the
    full_sums switch may change between computes (re-capture) without changing a bit of the results."""
    w, h = 200, 120
    nv = [nv12_frames(w, h, n) for n in range(3)]
    r8, d8 = tm.synth.rgb8_pair(w, h)
    rgb = (tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8))
    m = tm.Metrics(ssimulacra2=True, psnr=True, ssim=True)

    def run(eng, seq):
        out = []
        for frames, full in seq:
            eng.set_full_sums(full)
            for slot, (fr, fd) in enumerate(frames):
                eng.set_pair(slot, fr, fd)
            eng.compute_async(len(frames)); eng.sync()
            out.append([(eng.raw_sums(i).copy(), eng.scores(i), eng.sse(i)) for i in range(len(frames))])
        return out
    seq = [(nv, False), (nv[::-1], False), (nv[:2], False), ([rgb, rgb, rgb], False), (nv, True), ([nv[0], rgb], False), (nv, False)]
    a, b = tm.TurboMetrics(w, h, m, batch=3), tm.TurboMetrics(w, h, m, batch=3)
    b.set_graph(False)
    ra, rb = run(a, seq), run(b, seq)
    for x, y in zip(ra, rb):
        for (sa, ca, ea), (sb, cb, eb) in zip(x, y):
            assert np.array_equal(sa, sb) and ca == cb and ea == eb
    assert [s for s, _, _ in ra[0]][0].tolist() == [s for s, _, _ in ra[6]][0].tolist()   # replay after re-captures
    a.close(); b.close()


def test_psnr_and_ssim_without_ssimulacra2_skip_the_xyb_machinery():
    """PSNR / SSIM / MS-SSIM alone: no XYB pyramid is computed or allocated; the values equal those of the fused pass."""
    w, h = 640, 360
    fr, fd = nv12_frames(w, h, 4)
    fused = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=True, msssim=True), batch=1)
    light = tm.TurboMetrics(w, h, tm.Metrics(psnr=True, ssim=True, msssim=True), batch=1)
    a, b = fused.compute_one(fr, fd), light.compute_one(fr, fd)
    assert (a.psnr, a.ssim, a.msssim) == (b.psnr, b.ssim, b.msssim) and b.ssimulacra2 is None
    assert light.mem_usage() < fused.mem_usage() / 4
    with pytest.raises(tm.TmError):
        light.raw_sums(0)
    fused.close(); light.close()


def test_extreme_samples_match_oracle():
    """Out-of-range and degenerate samples: NV12 / P016 code values 0 and max (below black, above white, saturated chroma),
    flat frames, linear f32 input with negatives, zeros, huge values and denormals (the cube root's range fallback)."""
    w, h = 96, 80
    rng = np.random.default_rng(12)
    frames = []
    # NV12 with every code value, including the ones outside the limited range
    for seed in range(2):
        pl = lambda shape: rng.integers(0, 256, shape)
        ref = (pl((h, w)), pl((h // 2, w // 2)), pl((h // 2, w // 2)))
        dis = tuple(np.clip(p + rng.integers(-3, 4, p.shape), 0, 255) for p in ref)
        (rs, rp, rch), (ds, dp, dch) = tm.synth.pack_biplanar(ref, w, h, 8), tm.synth.pack_biplanar(dis, w, h, 8)
        frames.append((tm.HwFrame.nv12(rs, rp, rch, tm.ColorMatrix(seed)), tm.HwFrame.nv12(ds, dp, dch, tm.ColorMatrix(seed))))
    # P016 using all 16 bits (the reference kernel takes the full 16-bit value, biplanar.rs:89-101)
    ref = (rng.integers(0, 65536, (h, w)), rng.integers(0, 65536, (h // 2, w // 2)), rng.integers(0, 65536, (h // 2, w // 2)))
    dis = tuple(np.clip(p + rng.integers(-300, 301, p.shape), 0, 65535) for p in ref)
    (rs, rp, rch), (ds, dp, dch) = tm.synth.pack_biplanar(ref, w, h, 16), tm.synth.pack_biplanar(dis, w, h, 16)
    frames.append((tm.HwFrame.p016(rs, rp, rch), tm.HwFrame.p016(ds, dp, dch)))
    # flat black against flat white
    z, o = np.zeros((h, w, 3), np.uint8), np.full((h, w, 3), 255, np.uint8)
    frames.append((tm.HwFrame.rgb(z), tm.HwFrame.rgb(o)))
    # linear f32 far outside [0, 1]
    lf = (rng.standard_normal((h, w, 3)) * 2).astype(np.float32)
    lf[0, :8] = [0.0, -0.0, 1e-45, 1e-38, 1e30, 3e38, -1e30, 65504.0][:8][0]  # one row of specials, channel-wise below
    specials = np.array([0.0, -0.0, 1e-45, 1e-38, 1e30, 3e38, -1e30, 1e-20], np.float32)
    lf[1, :8, 0] = specials; lf[2, :8, 1] = specials; lf[3, :8, 2] = specials
    ld = (lf * np.float32(0.9)).astype(np.float32)
    frames.append((tm.HwFrame.linear(lf), tm.HwFrame.linear(ld)))
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=len(frames), full_sums=True)
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    eng.compute_async(); eng.sync()
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        sums, pyr = O.ssimulacra2_sums(lin[0], lin[1], want_xyb=True)
        for s in range(6):
            for side in range(2):
                for c in range(3):
                    got, want = eng.read_plane(slot, F.TM_PLANE_XYB, s, side, c), pyr[s][side][c]
                    assert np.array_equal(got, want, equal_nan=True), ("xyb", slot, s, side, c)
        got_s = eng.raw_sums(slot)
        ok = np.isfinite(sums)
        np.testing.assert_allclose(got_s[ok], sums[ok], rtol=1e-11, atol=1e-300)
        assert np.array_equal(np.isfinite(got_s), ok)
        if slot < 4:
            assert eng.sse(slot) == O.psnr(lin[0], lin[1])[0]
    eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_sweep_of_sizes_kinds_and_metric_masks(seed):
    """seeded random cases over what a caller can vary at once: frame size (incl. sizes around the tile / strip / segment borders
    of the kernels), input kind, colour matrix, metric mask, pruned or full sums, batch with different content per slot"""
    rng = np.random.default_rng(1000 + seed)
    edges = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 117, 118, 119, 127, 128, 129, 191, 192, 193, 202, 203, 255, 256, 257, 383, 384, 385]
    w = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 420))
    h = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 420))
    kind = str(rng.choice(["nv12", "p016", "rgb8", "rgb16", "rgbf32", "p10"]))  # p10 (round 6): 10-bit planes packed three samples to a word
    matrix = tm.ColorMatrix(int(rng.integers(0, 3)))
    want_ms = w >= 176 and h >= 176 and rng.random() < 0.7
    want_ssim = w >= 11 and h >= 11 and rng.random() < 0.7
    m = tm.Metrics(ssimulacra2=True, psnr=bool(rng.random() < 0.7), ssim=want_ssim, msssim=want_ms)
    full = bool(rng.random() < 0.4)
    B = int(rng.integers(1, 4))
    frames, packed = [], []
    for slot in range(B):
        n = int(rng.integers(0, 50))
        if kind == "nv12":
            frames.append(nv12_frames(w, h, n, matrix))
        elif kind == "p016":
            frames.append(p016_frames(w, h, n))
        elif kind == "p10":
            ref, dis = tm.synth.yuv420_pair(w, h, n, 10)
            packed.append(tuple(tm.HwFrame.i420p10(*(tm.synth.p10_pack_plane(p) for p in side), matrix=matrix) for side in (ref, dis)))
            frames.append(tuple(tm.HwFrame.p016(*tm.synth.pack_biplanar(side, w, h, 10), matrix) for side in (ref, dis)))  # what the oracle sees
        else:
            r8 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            d8 = np.clip(r8.astype(np.int32) + rng.integers(-9, 10, r8.shape), 0, 255).astype(np.uint8)
            if kind == "rgb8":
                frames.append((tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8)))
            elif kind == "rgb16":
                frames.append((tm.HwFrame.rgb(r8.astype(np.uint16) * 257), tm.HwFrame.rgb((d8.astype(np.uint16) * 257 + rng.integers(0, 99, d8.shape)).astype(np.uint16))))
            else:
                frames.append((tm.HwFrame.rgb(r8.astype(np.float32) / 255), tm.HwFrame.rgb(d8.astype(np.float32) / 255)))
    eng = tm.TurboMetrics(w, h, m, batch=B, full_sums=full)
    for slot, (fr, fd) in enumerate(packed or frames):
        eng.set_pair(slot, fr, fd)
    eng.compute_async()
    eng.sync()
    used = ssim_sums_used(ssim=want_ssim, msssim=want_ms) if (want_ssim or want_ms) else None
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        sums = O.ssimulacra2_sums(lin[0], lin[1])
        got = eng.raw_sums(slot)
        if full:
            np.testing.assert_allclose(got, sums, rtol=1e-12, atol=1e-300)
        else:  # the pruned default leaves the sums that carry no weight at 0
            nz = got != 0
            np.testing.assert_allclose(got[nz], np.asarray(sums)[nz], rtol=1e-12, atol=1e-300)
        s = eng.scores(slot)
        assert abs(s.ssimulacra2 - O.score_from_sums(sums, w, h)) <= 1e-9, (w, h, kind, slot)
        if m.psnr:
            sse, psnr = O.psnr(lin[0], lin[1])
            assert eng.sse(slot) == sse and s.psnr == psnr
        if used is not None:
            ws, wm, ssums = O.ssim_msssim(lin[0], lin[1])
            gs = eng.ssim_sums(slot)
            u = np.ones_like(used) if full else used
            nsc = 5 if want_ms else 1
            np.testing.assert_allclose(gs[:, :nsc][u[:, :nsc]], ssums[:, :nsc][u[:, :nsc]], rtol=1e-12, atol=1e-300)
            if want_ssim:
                assert abs(s.ssim - ws) <= 1e-6
            if want_ms:
                assert abs(s.msssim - wm) <= 1e-6
    eng.close()


# ---- round 4: the SHIPPING configuration against the oracle, directly (VERDICT r03 #2a) --------------------------------------
@pytest.mark.parametrize("kind,w,h,batch", [("nv12", 1920, 1080, 6), ("p016", 3840, 2160, 3)])
def test_shipping_configuration_against_the_oracle(kind, w, h, batch):
    """What bench.py and the CLI run: default variant, pruned sums, a launch large enough for k_blur_edge_fused to run as the
    persistent launch beside the two passes at raised priority (1080p from 5 pairs, 4K from 3).  Round 3 compared this
    configuration with the two-pass kernels at full size and with the oracle only up to 333 x 203; here every slot's weighted
    sums and score are held to the oracle itself (the oracle takes ~4 s per 1080p pair, ~16 s per 4K pair)."""
    mk = nv12_frames if kind == "nv12" else p016_frames
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=batch)
    frames = [mk(w, h, 40 + n) for n in range(batch)]
    for slot, (fr, fd) in enumerate(frames):
        eng.set_pair(slot, fr, fd)
    assert eng.uses_fused_edge(batch) and not eng.uses_fused_edge(1)  # the engine's own choice, nothing forced
    for rep in range(2):  # a second launch finds the hand-off words of the first in place
        eng.compute_async(batch)
        eng.sync()
    m = weight_mask()
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        want, sums = O.ssimulacra2_from_linear(lin[0], lin[1])
        got = eng.raw_sums(slot)
        np.testing.assert_allclose(got[m], np.asarray(sums).reshape(6, 6, 3)[m], rtol=1e-12, atol=1e-300, err_msg=f"slot {slot}")
        modes = eng.job_modes()  # pruned by job: a (scale, channel) image without any weight is not computed at all, an edge-only one has no ssim sums
        for sc in range(6):
            for c in range(3):
                if modes[sc, c] == 0:
                    assert np.all(got[sc, :, c] == 0.0)
                elif modes[sc, c] == 1:
                    assert got[sc, 0, c] == 0.0 and got[sc, 3, c] == 0.0
        assert (modes == 1).sum() == 2 and (modes == 0).sum() >= 1
        assert abs(eng.scores(slot).ssimulacra2 - want) <= 1e-9, slot
    eng.close()


@pytest.mark.parametrize("graph", [False, True])
def test_fused_edge_handoff_timeout_is_reported_and_the_engine_recovers(graph):
    """VERDICT r03 #2b: the failure path of the chained bands.  With TM_DBG_EF_FAULT = 2 the fused kernel does not publish the
    column state between groups of four bands: the next group's wait gives up, tm_engine_sync returns TM_ERR_HIP, the results of
    that launch are not available -- and the very next launch (fault off) delivers the correct sums again, directly and under
    hipGraph replay (reference: a failed CUDA call surfaces as Err from compute_sync, ssimulacra2-cuda/src/lib.rs:271-291)."""
    w, h, batch = 640, 360, 3  # 12 bands of 32 rows = 3 groups: two boundaries cross memory
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=batch)
    for slot in range(batch):
        eng.set_pair(slot, *nv12_frames(w, h, 7 + slot))
    eng.set_variant(F.TM_VARIANT_TWO_PASS_EDGE)
    eng.compute_async(); eng.sync()
    want = [eng.raw_sums(i).copy() for i in range(batch)]
    eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    eng.set_graph(graph)
    eng.compute_async(); eng.sync()
    assert all(np.array_equal(eng.raw_sums(i), want[i]) for i in range(batch))
    for round_ in range(2):
        eng.debug_set_param(F.TM_DBG_EF_FAULT, 2)
        eng.compute_async()
        with pytest.raises(tm.TmError) as ei:
            eng.sync()
        assert ei.value.code == F.TM_ERR_HIP and "hand-off" in str(ei.value)
        with pytest.raises(tm.TmError) as ei:  # no results of the failed launch
            eng.raw_sums(0)
        assert ei.value.code == F.TM_ERR_STATE
        eng.debug_set_param(F.TM_DBG_EF_FAULT, 0)
        for _ in range(2):
            eng.compute_async(); eng.sync()
            assert all(np.array_equal(eng.raw_sums(i), want[i]) for i in range(batch)), round_
    with pytest.raises(tm.TmError):
        eng.debug_set_param(F.TM_DBG_EF_FAULT, 9)
    with pytest.raises(tm.TmError):
        eng.debug_set_param(99, 0)
    eng.close()


def test_stale_handoff_tags_cannot_match_after_the_epoch_wraps():
    """the hand-off tags carry 24 bits of the launch epoch, and the words of slots that later launches do not touch keep
    their tags.  Launch 3 slots at epoch 5; wrap the epoch with 1-slot launches (0xFFFFFE, 0xFFFFFF, 1, 2, 3, 4); launch 3 slots
    again at epoch 5 with OTHER frames in slots 1, 2: a stale word of the first launch carries exactly the tag a reader expects.
    The launch that finds the epoch at 1 clears the words, so the readers wait for the real producers."""
    w, h = 640, 360
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=3)
    eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    a = [nv12_frames(w, h, 20 + n) for n in range(3)]
    b = [a[0]] + [nv12_frames(w, h, 30 + n) for n in range(1, 3)]
    for slot, (fr, fd) in enumerate(a):
        eng.set_pair(slot, fr, fd)
    eng.debug_set_edge_epoch(5)
    eng.compute_async(3); eng.sync()
    first = [eng.raw_sums(i).copy() for i in range(3)]
    eng.debug_set_edge_epoch(0xFFFFFE)
    for _ in range(6):  # epochs ..FE, ..FF, 1 (clears), 2, 3, 4
        eng.compute_async(1); eng.sync()
        assert np.array_equal(eng.raw_sums(0), first[0])
    for slot, (fr, fd) in enumerate(b):
        eng.set_pair(slot, fr, fd)
    eng.compute_async(3); eng.sync()  # epoch 5 again
    got = [eng.raw_sums(i).copy() for i in range(3)]
    eng.set_variant(F.TM_VARIANT_TWO_PASS_EDGE)
    eng.compute_async(3); eng.sync()
    for i in range(3):
        assert np.array_equal(got[i], eng.raw_sums(i)), i
    assert not np.array_equal(got[1], first[1])
    eng.close()


def test_declared_surfaces_and_separate_planes_give_the_same_bits():
    """tm_engine_set_surface_nv12 (ONE allocation: coded_height luma rows, padding included, then the CbCr rows -- the reference's
    from_mapping contract, one 2-D copy) against tm_engine_set_frame_nv12 on two planes that live in SEPARATE host allocations
    (copied plane by plane; round 3 guessed "one allocation" from the distance of the two pointers), padding rows poisoned."""
    import ctypes as C
    w, h = 150, 70
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 3)
    assert rch > h  # the synthetic surface has padding rows between the planes
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
    eng.set_pair(0, tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
    keep = []
    for side, (buf, pitch, ch) in enumerate(((rs, rp, rch), (ds, dp, dch))):
        a = np.asarray(buf, np.uint8)
        y = np.ascontiguousarray(a[: pitch * h]).copy()
        uv = np.ascontiguousarray(a[pitch * ch: pitch * ch + pitch * ((h + 1) // 2)]).copy()
        keep += [y, uv]
        tm.engine._chk(eng._L.tm_engine_set_frame_nv12(eng._h, 1, side, y.ctypes.data, uv.ctypes.data, pitch, 0, 0, 0, F.TM_MEM_HOST), "set_frame_nv12")
    eng.compute_async(2); eng.sync()
    assert np.array_equal(eng.raw_sums(0), eng.raw_sums(1)) and eng.sse(0) == eng.sse(1)
    # a coded height below the picture height is refused
    with pytest.raises(tm.TmError):
        eng.set_frame(0, 0, tm.HwFrame.nv12(rs, rp, h - 1))
    eng.close()


def test_toggling_full_sums_does_not_grow_the_engine():
    """make_job_tables lost track of the fused kernel's buffers when full_sums was switched on and off"""
    eng = tm.TurboMetrics(640, 360, tm.Metrics(ssimulacra2=True), batch=4)
    base = eng.mem_usage()
    for _ in range(4):
        eng.set_full_sums(True)
        assert not eng.uses_fused_edge()
        eng.set_full_sums(False)
    assert eng.mem_usage() == base
    eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    assert eng.uses_fused_edge()
    eng.close()


def test_upload_fences_say_when_a_page_locked_frame_may_be_overwritten():
    """tm_engine_upload_fence / tm_engine_upload_done: a TM_MEM_HOST_PINNED frame is the caller's again as soon as its DMA into the
    engine's device surface is done -- the host buffer is overwritten right after the fence has been waited for, BEFORE compute_async,
    and the scores are those of the bytes that were there at set_frame time (what lets the CLI recycle a ring of five page-locked
    pictures per stream whatever the batch size)"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    w, h = 640, 360
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2)
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 4)
    eng.set_pair(0, tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))  # pageable reference run
    eng.compute_async(1); eng.sync()
    want = eng.raw_sums(0).copy(); want_sse = eng.sse(0)
    pr, pd = torch.from_numpy(np.asarray(rs).copy()).pin_memory(), torch.from_numpy(np.asarray(ds).copy()).pin_memory()
    L, hnd = eng._L, eng._h
    tok = C.c_uint64()
    assert L.tm_engine_upload_done(hnd, 0, 0) < 0  # no fence taken yet: invalid token
    for slot in range(2):
        eng.set_pair(slot, tm.HwFrame.nv12(pr, rp, rch), tm.HwFrame.nv12(pd, dp, dch))
        assert L.tm_engine_upload_fence(hnd, C.byref(tok)) == 0 and tok.value == 2 * slot
        assert L.tm_engine_upload_done(hnd, tok.value, 1) == 1  # blocks until the two copies have left host memory
        assert L.tm_engine_upload_done(hnd, tok.value, 0) == 1
        assert eng.upload_done(eng.upload_fence(), block=True)  # the Python mirror of the same two calls
        pr.fill_(0); pd.fill_(255)                               # the host buffers are ours again ...
        if slot == 0:
            pr.copy_(torch.from_numpy(np.asarray(rs))); pd.copy_(torch.from_numpy(np.asarray(ds)))  # ... refilled for the next slot
    eng.compute_async(2); eng.sync()
    for slot in range(2):
        assert np.array_equal(eng.raw_sums(slot), want) and eng.sse(slot) == want_sse, slot
    # tokens older than the ring of 256 fence events are answered by the events that took their place (a later fence of the same
    # stream; the most recent fence of the second upload stream): done, and still valid -- never an error, never early
    first = eng.upload_fence()
    for k in range(300):
        eng.set_pair(k & 1, tm.HwFrame.nv12(pr, rp, rch), tm.HwFrame.nv12(pd, dp, dch))
        last = eng.upload_fence()
    assert last - first == 300
    assert L.tm_engine_upload_done(hnd, first, 1) == 1 and L.tm_engine_upload_done(hnd, first, 0) == 1
    assert L.tm_engine_upload_done(hnd, last, 1) == 1
    assert L.tm_engine_upload_done(hnd, last + 1, 0) < 0  # not handed out yet
    free, total = C.c_size_t(), C.c_size_t()
    assert L.tm_device_mem_info(C.byref(free), C.byref(total)) == 0 and 0 < free.value <= total.value and total.value > (64 << 30)
    eng.close()


def test_a_fence_covers_second_stream_copies_from_before_an_earlier_fence():
    """a fence stands for EVERY upload enqueued so far.  Distorted-side page-locked frames go up on the second upload
    stream; a later fence that only saw reference-side copies must still not be done before those (nothing on the engine's stream
    waits for them until the next launch).  4K P016 frames (25 MB, ~0.5 ms on the link each) make an early answer observable: once
    the newest fence is done, every older fence is done too."""
    torch = pytest.importorskip("torch")
    w, h = 3840, 2160
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=4)
    (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, 1)
    pr = torch.from_numpy(np.asarray(rs).copy()).pin_memory()
    pd = torch.from_numpy(np.asarray(ds).copy()).pin_memory()
    for rounds in range(8):
        older = []
        for slot in range(4):  # four 25-MB copies queued on the second stream, a fence after each
            eng.set_frame(slot, F.TM_SIDE_DIS, tm.HwFrame.p016(pd, dp, dch))
            older.append(eng.upload_fence())
        eng.set_frame(0, F.TM_SIDE_REF, tm.HwFrame.p016(pr, rp, rch))  # one copy on the engine's own stream
        newest = eng.upload_fence()
        assert eng.upload_done(newest, block=True)
        assert all(eng.upload_done(t, block=False) for t in older), rounds
    # and a distorted-side frame set again from pageable memory (engine's stream) lands AFTER the page-locked one (second stream)
    one = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    want = one.compute_one(tm.HwFrame.p016(rs, rp, rch), tm.HwFrame.p016(ds, dp, dch)).ssimulacra2
    for _ in range(4):
        eng.set_frame(0, F.TM_SIDE_REF, tm.HwFrame.p016(pr, rp, rch))
        eng.set_frame(0, F.TM_SIDE_DIS, tm.HwFrame.p016(pr, rp, rch))   # page-locked, second stream: the WRONG picture ...
        eng.set_frame(0, F.TM_SIDE_DIS, tm.HwFrame.p016(ds, dp, dch))   # ... replaced from pageable memory on the engine's stream
        eng.compute_async(1); eng.sync()
        assert eng.scores(0).ssimulacra2 == want
    eng.close(); one.close()


def test_compute_one_deferred_gives_compute_ones_scores_with_two_pairs_in_flight():
    """compute_one without its blocking sync -- a ticket per pair, collect(ticket) one call later (the three-line change
    to the reference's loop).  Same FrameScores as compute_one, bit for bit, whatever the collection order."""
    w, h = 640, 360
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=True), batch=1)
    frames = [nv12_frames(w, h, i) for i in range(5)]
    want = [eng.compute_one(fr, fd) for fr, fd in frames]
    assert len({s.ssimulacra2 for s in want}) == 5
    mem_one = eng.mem_usage()
    got, last = [], None
    for k in range(15):  # the reference's loop with a lag of one: submit pair k, then collect pair k - 1
        t = eng.compute_one_deferred(*frames[k % 5])
        if last is not None:
            got.append(eng.collect(last))
        last = t
    got.append(eng.collect(last))
    assert got == want * 3
    tickets = [eng.compute_one_deferred(*frames[k]) for k in range(5)]  # nothing collected in between: older pairs are finished and kept
    assert [eng.collect(t) for t in reversed(tickets)] == list(reversed(want))
    with pytest.raises(tm.TmError):
        eng.collect(tickets[0])  # a ticket is good for one collect
    with pytest.raises(tm.TmError):
        eng.collect(10 ** 6)
    assert eng.compute_one(*frames[2]) == want[2]  # the blocking call still works beside it
    t0, t1 = eng.compute_one_deferred(*frames[0]), eng.compute_one_deferred(*frames[1])
    assert eng.compute_one(*frames[3]) == want[3]  # ... also with two pairs in flight: they are finished first and kept
    assert eng.collect(t1) == want[1] and eng.collect(t0) == want[0]
    assert eng.mem_usage() == mem_one  # (the second engine is its own object)
    # settings are the engine's: the second engine follows them, whether they were made before or after it came to be
    eng.set_channel_mode(True)
    first = [eng.compute_one(*frames[k]) for k in range(3)]
    assert first[0].psnr != want[0].psnr and first[0].ssimulacra2 == want[0].ssimulacra2
    tk = [eng.compute_one_deferred(*frames[k]) for k in range(3)]
    assert [eng.collect(t) for t in tk] == first
    fresh = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True, ssim=True), batch=1)
    fresh.set_channel_mode(True); fresh.set_full_sums(True)
    tk = [fresh.compute_one_deferred(*frames[k]) for k in range(3)]
    assert [fresh.collect(t) for t in tk] == first
    fresh.close()
    batched = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=4)
    with pytest.raises(ValueError):
        batched.compute_one_deferred(*frames[0])
    batched.close(); eng.close()


def test_a_setting_changed_between_submit_and_collect_keeps_the_pairs_in_flight():
    """tm_engine_set_full_sums drops the engine's results, so a setter that ran before the deferred pairs were retired lost
    their scores and wedged every later collect.  Setters retire first: a pair in flight is scored under the settings it was submitted
    with, and the engines then take the new one."""
    w, h = 416, 240
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=1)
    frames = [nv12_frames(w, h, 20 + i) for i in range(4)]
    want = [eng.compute_one(fr, fd) for fr, fd in frames]
    t0, t1 = eng.compute_one_deferred(*frames[0]), eng.compute_one_deferred(*frames[1])
    eng.set_full_sums(True)  # both engines have a pair in flight
    assert eng.collect(t0) == want[0] and eng.collect(t1) == want[1]
    t2 = eng.compute_one_deferred(*frames[2])
    eng.set_channel_mode(True)  # the pair in flight is scored pooled, the next one from channel 0
    t3 = eng.compute_one_deferred(*frames[3])
    assert eng.collect(t2) == want[2]
    got3 = eng.collect(t3)
    assert got3.ssimulacra2 == want[3].ssimulacra2 and got3.psnr != want[3].psnr
    t = eng.compute_one_deferred(*frames[0])
    eng.set_full_sums(False); eng.set_channel_mode(False)
    assert eng.collect(t).ssimulacra2 == want[0].ssimulacra2
    assert eng.compute_one(*frames[1]) == want[1]
    eng.close()


def test_the_first_launch_after_switching_to_the_reference_pipeline_is_correct():
    """Round 5 (found by tests/soak/variant_sweep_soak.py): TM_VARIANT_REFERENCE allocates its linear pyramid and transposed XYB copy when it
    is first selected; their zero fill ran on the null stream and could still be running when the first launch's kernels -- on the engine's
    non-blocking stream -- wrote into them: at 1080p x 4 the first launch after the switch returned garbage, the second one was right."""
    w, h, B = 1920, 1080, 4
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B)
    for slot in range(B):
        eng.set_pair(slot, *nv12_frames(w, h, 60 + slot))
    eng.compute_async(); eng.sync()
    want = [(eng.raw_sums(i).copy(), eng.sse(i)) for i in range(B)]
    eng.set_variant(F.TM_VARIANT_REFERENCE)
    eng.compute_async(); eng.sync()  # the FIRST launch on the freshly allocated arenas
    for i in range(B):
        assert np.array_equal(eng.raw_sums(i), want[i][0]) and eng.sse(i) == want[i][1], i
    eng.set_full_sums(True)  # (may grow the fused kernel's hand-off buffers: the same zero fill)
    eng.set_variant(F.TM_VARIANT_FUSED_EDGE)
    eng.compute_async(); eng.sync()
    m = weight_mask()
    for i in range(B):
        assert np.array_equal(eng.raw_sums(i)[m], want[i][0][m]), i
    eng.close()


def test_destroying_a_chained_peer_first_unhooks_it():
    w, h = 320, 200
    a = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    b = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    fr, fd = nv12_frames(w, h, 2)
    want = a.compute_one(fr, fd).ssimulacra2
    a.set_graph(False)
    a.debug_chain(b)
    assert a.compute_one(fr, fd).ssimulacra2 == want
    b.close()  # a's launches waited on b's events
    assert a.compute_one(fr, fd).ssimulacra2 == want
    a.close()


@pytest.mark.parametrize("w,h", [(16384, 40), (40, 16384)])
def test_maximum_width_and_height_against_the_oracle(w, h):
    """the largest picture sides tm_engine_create accepts (16 384): 512 tiles / 512 bands in the fused kernel, 256 column blocks / 256 row
    blocks in the two passes, byte offsets close to 2^32 inside a plane -- every weighted sum against the oracle, default kernels, the
    fused kernel forced, and the two-pass kernels, bit for bit among themselves"""
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2)
    fr, fd = nv12_frames(w, h, 9)
    eng.set_pair(0, fr, fd); eng.set_pair(1, fd, fr)
    got = {}
    for name, variant in (("default", F.TM_VARIANT_DEFAULT), ("fused", F.TM_VARIANT_FUSED_EDGE), ("two_pass", F.TM_VARIANT_TWO_PASS_EDGE),
                          ("whole_rows", F.TM_VARIANT_WHOLE_ROWS | F.TM_VARIANT_TWO_PASS_EDGE)):
        eng.set_variant(variant)
        eng.compute_async(); eng.sync()
        got[name] = [eng.raw_sums(i).copy() for i in range(2)]
    for name in got:
        assert all(np.array_equal(got[name][i], got["two_pass"][i]) for i in range(2)), name
    lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
    want, sums = O.ssimulacra2_from_linear(lin[0], lin[1])
    m = weight_mask()
    np.testing.assert_allclose(got["default"][0][m], np.asarray(sums).reshape(6, 6, 3)[m], rtol=1e-12, atol=1e-300)
    assert abs(eng.scores(0).ssimulacra2 - want) <= 1e-9
    sse, psnr = O.psnr(lin[0], lin[1])
    assert eng.sse(0) == sse == eng.sse(1) and eng.scores(0).psnr == psnr
    eng.close()
    with pytest.raises(tm.TmError):
        tm.TurboMetrics(16385, 16, tm.Metrics(ssimulacra2=True), batch=1)


def test_many_slots_of_small_frames():
    """256 slots of 96 x 64 frames in one launch (slot-major grids of 256 x jobs workgroups, 512 planes for the fused kernel): every slot's
    sums equal those of the same pair computed alone"""
    w, h, B = 96, 64, 256
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
    one = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    frames = [nv12_frames(w, h, n) for n in range(5)]
    for slot in range(B):
        eng.set_pair(slot, *frames[slot % 5])
    alone = []
    for fr, fd in frames:
        one.compute_one(fr, fd)
        alone.append(one.raw_sums(0).copy())
    for variant in (F.TM_VARIANT_DEFAULT, F.TM_VARIANT_FUSED_EDGE):
        eng.set_variant(variant)
        eng.compute_async(B); eng.sync()
        assert eng.uses_fused_edge(B)
        for slot in range(B):
            assert np.array_equal(eng.raw_sums(slot), alone[slot % 5]), (variant, slot)
    eng.close(); one.close()


@pytest.mark.parametrize("bits,w,h", [(8, 1920, 1080), (10, 640, 360), (8, 328, 201), (8, 70, 38), (10, 333, 203), (12, 136, 22)])
def test_a_tight_planar_picture_goes_up_as_one_copy_with_the_same_results(bits, w, h):
    """a picture as it lies in a Y4M / raw planar file -- rows without padding, Cb behind Y, Cr behind Cb -- in host memory: ONE linear copy,
    read by the kernels with the file's own pitches (chroma rows of 4n bytes; the others take the 2-D copies), against the same three
    planes from separate arrays (2-D copies into padded rows) and with TM_DBG_LINEAR_UPLOAD = 0: raw sums, SSE and scores bit for bit;
    pageable and page-locked memory"""
    import torch
    dt = np.uint8 if bits == 8 else np.uint16
    cw, ch = (w + 1) // 2, (h + 1) // 2
    pairs = []
    for n in range(2):
        ref, dis = tm.synth.yuv420_pair(w, h, n + 3, 8 if bits == 8 else 10)
        if bits == 12:
            ref, dis = tuple(p * 4 + 3 for p in ref), tuple(p * 4 + 1 for p in dis)
        pairs.append((ref, dis))

    def tight(planes, pinned):
        flat = np.concatenate([np.ascontiguousarray(p.astype(dt)).reshape(-1) for p in planes])
        if pinned:
            t = torch.from_numpy(flat).pin_memory()
            return t[:w * h].view(h, w), t[w * h:w * h + cw * ch].view(ch, cw), t[w * h + cw * ch:].view(ch, cw)
        return flat[:w * h].reshape(h, w), flat[w * h:w * h + cw * ch].reshape(ch, cw), flat[w * h + cw * ch:].reshape(ch, cw)

    got = {}
    for name, linear, make in (("separate", 1, lambda pl, slot: [np.ascontiguousarray(p.astype(dt)) for p in pl]),
                               ("tight", 1, lambda pl, slot: tight(pl, slot == 1)),
                               ("tight_2d", 0, lambda pl, slot: tight(pl, slot == 1)),
                               ("tight_one_stream", 2, lambda pl, slot: tight(pl, slot == 1))):
        eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=2, full_sums=True)
        eng.debug_set_param(F.TM_DBG_LINEAR_UPLOAD, 1 if linear else 0)
        eng.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, 1 if linear == 2 else 2)
        for rep in range(2):  # the second round over the same staging surfaces
            for slot, (ref, dis) in enumerate(pairs):
                for side, planes in enumerate((ref, dis) if rep == 0 else (dis, ref)):
                    pl = make(planes, slot)
                    eng.set_frame(slot, side, tm.HwFrame.i420(pl[0], pl[1], pl[2], bits=bits))
            eng.compute_async(); eng.sync()
            got[(name, rep)] = [(eng.raw_sums(s).copy(), eng.sse(s), eng.scores(s)) for s in range(2)]
        eng.close()
    for rep in range(2):
        for name in ("tight", "tight_2d", "tight_one_stream"):
            for s in range(2):
                a, b = got[(name, rep)][s], got[("separate", rep)][s]
                assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2], (name, rep, s)
    assert not np.array_equal(got[("tight", 0)][0][0], got[("tight", 0)][1][0])


@pytest.mark.parametrize("streams", [2, 1])
def test_page_locked_frames_on_two_upload_streams_keep_their_order_with_the_launches(streams):
    """page-locked frames of the distorted side go up on the device's second upload stream: 40 launches on two engines that take turns
    (the CLI's ping-pong), every launch with other frames in the same staging surfaces and no host synchronisation between handing the
    frames over and launching -- a copy that overtook the launch before it, or a launch that did not wait for its copies, shows up as the
    sums of the wrong frames.  Fences: a frame's bytes may be overwritten once its fence is done, and the results still come out right."""
    import torch
    w, h, B = 640, 360, 4
    engs = [tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=B) for _ in range(2)]
    for e in engs:
        e.debug_set_param(F.TM_DBG_UPLOAD_STREAMS, streams)
    one = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True, psnr=True), batch=1)
    src = []  # six different pairs, as NV12 surfaces and as tight planar pictures
    want = []
    for n in range(6):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        one.compute_one(tm.HwFrame.nv12(rs, rp, rch), tm.HwFrame.nv12(ds, dp, dch))
        want.append((one.raw_sums(0).copy(), one.sse(0)))
        src.append(((rs, rp, rch), (ds, dp, dch)))
    ring = [[torch.empty_like(torch.from_numpy(src[0][side][0])).pin_memory() for side in range(2)] for _ in range(2 * B + 2)]
    fences = [None] * len(ring)
    k = 0
    pending = [None, None]

    def check(i):
        if pending[i] is None:
            return
        engs[i].sync()
        for slot, n in enumerate(pending[i]):
            assert np.array_equal(engs[i].raw_sums(slot), want[n][0]) and engs[i].sse(slot) == want[n][1], (streams, i, slot, n)
        pending[i] = None

    for launch in range(40):
        i = launch & 1
        check(i)
        picks = [(launch * 5 + slot * 7) % 6 for slot in range(B)]
        for slot, n in enumerate(picks):
            r = k % len(ring); k += 1
            if fences[r] is not None:
                assert fences[r][0].upload_done(fences[r][1], block=True)
            for side in range(2):
                ring[r][side].copy_(torch.from_numpy(src[n][side][0]))
            engs[i].set_pair(slot, tm.HwFrame.nv12(ring[r][0], src[n][0][1], src[n][0][2]), tm.HwFrame.nv12(ring[r][1], src[n][1][1], src[n][1][2]))
            fences[r] = (engs[i], engs[i].upload_fence())
        engs[i].compute_async(B)
        pending[i] = picks
    check(0); check(1)
    for e in engs + [one]:
        e.close()


def test_device_numa_node_is_a_node_of_this_host_or_unknown():
    """tm_device_numa_node: the host NUMA node next to the device (what the CLI binds its threads to), -1 when unknown or for a device
    that does not exist"""
    L = tm.ffi.lib()
    node = L.tm_device_numa_node(0)
    assert node == -1 or os.path.isdir(f"/sys/devices/system/node/node{node}"), node
    assert L.tm_device_numa_node(4096) == -1 and L.tm_device_numa_node(-1) == -1
