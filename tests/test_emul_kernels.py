"""No-GPU tier: execute the product's kernel source lane by lane on the CPU (tests/emul) and demand
bit equality with the oracle for every plane of every scale, and 1e-12 agreement of the sums.
The same comparisons run against the real device in tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.emul import emul as E
from tm_pkg import tm


def coef_table():
    return np.stack([np.stack([O.yuv_coefficients(m, 8), O.yuv_coefficients(m, 16)]) for m in range(3)]).astype(np.float32)


def oracle_linear(f, w, h):
    k = f["kind"]
    if k == "i420p10":  # the packed upload kind: unpacked by the numpy statement of the layout, then like planar 10-bit
        cw, ch = (w + 1) // 2, (h + 1) // 2
        planes = tuple(tm.synth.p10_unpack_plane(p, n) for p, n in zip(f["data"], (w, cw, cw)))
        surf, pitch, coded = tm.synth.pack_biplanar(planes, w, h, 10)
        return O.yuv420_biplanar_to_linear(surf, pitch, coded, w, h, 16, int(f.get("matrix", 0)))
    if k == "i420":  # planar: the oracle sees the same samples repacked into the reference's biplanar surface
        surf, pitch, ch = tm.synth.pack_biplanar(tuple(np.asarray(p, np.int64) for p in f["data"]), w, h, f["bits"])
        return O.yuv420_biplanar_to_linear(surf, pitch, ch, w, h, 8 if f["bits"] == 8 else 16, int(f.get("matrix", 0)))
    if k in ("nv12", "p016"):
        return O.yuv420_biplanar_to_linear(f["data"], f["pitch"], f["coded_height"], w, h, 8 if k == "nv12" else 16, int(f.get("matrix", 0)))
    return {"rgb8": O.rgb8_to_linear, "rgb16": O.rgb16_to_linear, "rgbf32": O.rgbf32_to_linear, "linear_f32": O.linear_packed_to_planar}[k](f["data"])


def check_against_oracle(em, frames, w, h, have_linear=True, have_xybt=True):
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        sums, pyr = O.ssimulacra2_sums(lin[0], lin[1], want_xyb=True)
        for side in range(2):
            for c in range(3):
                if have_linear:
                    assert np.array_equal(em.plane(em.LIN, slot, 0, side, c), lin[side][c]), ("linear", slot, side, c)
        for s, (ws, hs) in enumerate(O.scale_sizes(w, h)):
            for side in range(2):
                for c in range(3):
                    got = em.plane(em.XYB, slot, s, side, c)
                    assert np.array_equal(got, pyr[s][side][c]), ("xyb", slot, s, side, c)
                    if have_xybt:
                        got_t = em.plane(em.XYBT, slot, s, side, c, transposed=True)
                        assert np.array_equal(got_t, pyr[s][side][c].T), ("xybt", slot, s, side, c)
            _, cap = O.process_scale(pyr[s][0], pyr[s][1], capture=True)
            for p in range(5):
                for c in range(3):
                    got = em.plane(em.V, slot, s, p, c, transposed=True, per_slot=5)
                    assert np.array_equal(got, cap["pass1"][p][c]), ("pass1", slot, s, p, c)
        np.testing.assert_allclose(em.sums(slot), sums, rtol=1e-12, atol=1e-300)
        sse, _ = O.psnr(lin[0], lin[1])
        assert int(em.SSE[slot]) == sse


REFERENCE, DEFAULT, WIDE_ROWS, TILE_INGEST, SPLIT_ROWS = 1, 0, 0x100, 0x200, 0x400  # emul_pipeline variants == TM_VARIANT_* of the engine


def nv12_frames(w, h, count=2):
    frames = []
    for n in range(count):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        frames.append((dict(kind="nv12", data=rs, pitch=rp, coded_height=rch, matrix=n % 3),
                       dict(kind="nv12", data=ds, pitch=dp, coded_height=dch, matrix=n % 3)))
    return frames


def weight_mask():
    """(6 scales, 6 kinds, 3 channels) bool: sums that carry a non-zero weight (table layout [channel][scale][kind])"""
    return (O.weights().reshape(3, 6, 6) != 0.0).transpose(1, 2, 0)


@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (64, 64), (1, 1), (2, 5), (129, 20)])
def test_reference_pipeline_matches_oracle(w, h):
    """the straight-line kernels (k_ingest, k_downscale, k_xyb, k_blur_v, k_blur_h_jobs): every plane incl. linear RGB and the
    transposed XYB copy"""
    frames = nv12_frames(w, h)
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=REFERENCE)
    check_against_oracle(em, frames, w, h)


@pytest.mark.parametrize("variant", [REFERENCE, DEFAULT])
def test_p016_and_rgb_kinds_match_oracle(variant):
    w, h = 46, 30
    (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, 3)
    r8, d8 = tm.synth.rgb8_pair(w, h)
    rng = np.random.default_rng(5)
    r16 = (r8.astype(np.uint16) * 257); d16 = (d8.astype(np.uint16) * 257 + rng.integers(0, 200, d8.shape)).astype(np.uint16)
    rf = rng.random((h, w, 3), dtype=np.float32); df = np.clip(rf + rng.normal(0, 0.03, rf.shape).astype(np.float32), 0, 1)
    frames = [
        (dict(kind="p016", data=rs, pitch=rp, coded_height=rch, matrix=0), dict(kind="p016", data=ds, pitch=dp, coded_height=dch, matrix=0)),
        (dict(kind="rgb8", data=r8), dict(kind="rgb8", data=d8)),
        (dict(kind="rgb16", data=r16), dict(kind="rgb16", data=d16)),
        (dict(kind="rgbf32", data=rf), dict(kind="rgbf32", data=df)),
        (dict(kind="linear_f32", data=rf), dict(kind="linear_f32", data=df)),
    ]
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant)
    check_against_oracle(em, frames, w, h, have_linear=variant == REFERENCE, have_xybt=variant == REFERENCE)


def test_wave_ingest_p016_launch_and_unaligned_pitch():
    """all-P016 launches of the wave ingest (KIND-specialised kernel, pair loads): slot 0 = 10-bit samples in the upper bits,
    slot 1 = full 16-bit content, slot 2 = a BT.601 matrix; NV12 with an odd pitch goes through the unaligned branch of the pair loads"""
    w, h = 46, 30
    frames = []
    for n, matrix in ((3, 0), (4, 0), (5, 2)):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, n)
        if n == 4:
            rs = rs.copy(); ds = ds.copy()
            rs.view(np.uint16)[::7] |= 0x21; ds.view(np.uint16)[::5] |= 0x3F  # 16-bit content
        frames.append((dict(kind="p016", data=rs, pitch=rp, coded_height=rch, matrix=matrix), dict(kind="p016", data=ds, pitch=dp, coded_height=dch, matrix=matrix)))
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
    # NV12, pitch not a multiple of 2
    frames = []
    for n in range(2):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
        def repitch(buf, pitch, rows, newp):
            a = np.asarray(buf, np.uint8)[: pitch * rows].reshape(rows, pitch)
            out = np.zeros((rows, newp), np.uint8); out[:, : min(pitch, newp)] = a[:, : min(pitch, newp)]
            return out.reshape(-1).copy()
        rows = rch * 3 // 2
        newp = w + 1 if (w + 1) % 2 else w + 3
        frames.append((dict(kind="nv12", data=repitch(rs, rp, rows, newp), pitch=newp, coded_height=rch, matrix=1),
                       dict(kind="nv12", data=repitch(ds, dp, rows, newp), pitch=newp, coded_height=dch, matrix=1)))
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)


def test_default_pipeline_on_random_frame_sizes():
    """twenty seeded random sizes (1 .. 300 pixels each way, mostly odd) through the default pipeline, two slots each"""
    rng = np.random.default_rng(20250101)
    for _ in range(20):
        w, h = int(rng.integers(1, 301)), int(rng.integers(1, 301))
        frames = []
        for n in range(2):
            (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, int(rng.integers(0, 50)))
            frames.append((dict(kind="nv12", data=rs, pitch=rp, coded_height=rch, matrix=int(rng.integers(0, 3))),
                           dict(kind="nv12", data=ds, pitch=dp, coded_height=dch, matrix=0)))
            frames[-1][1]["matrix"] = frames[-1][0]["matrix"]
        em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
        check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)


def test_default_pipeline_on_a_multi_tile_frame():
    """640x360 through the default pipeline: 20 x 45 ingest tiles, 10 column blocks, 6 row blocks at scale 0, every
    scale with partial tiles somewhere; planes bit-exact, pruned sums and score equal to the full computation"""
    w, h = 640, 360
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 7)
    frames = [(dict(kind="nv12", data=rs, pitch=rp, coded_height=rch, matrix=0), dict(kind="nv12", data=ds, pitch=dp, coded_height=dch, matrix=0))]
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
    pr = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=False)
    m = weight_mask()
    assert np.array_equal(pr.sums(0).reshape(6, 6, 3)[m], em.sums(0).reshape(6, 6, 3)[m])


@pytest.mark.parametrize("variant", [REFERENCE, DEFAULT])
@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (1, 1), (129, 20), (200, 9)])
def test_job_driven_blur_passes_full_and_pruned(w, h, variant):
    """full_sums: every plane and all 108 sums equal the oracle.  Pruned (the default of the engine): the sums that carry
    weight are bit-identical to the full run, the others read 0, and the score is bit-identical."""
    frames = nv12_frames(w, h)
    full = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=True)
    check_against_oracle(full, frames, w, h, have_linear=False, have_xybt=variant == REFERENCE)
    pruned = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=False)
    m = weight_mask()
    assert m.sum() == 52
    for slot in range(2):
        a, b = full.sums(slot), pruned.sums(slot)
        assert np.array_equal(a[m], b[m])
        assert np.all(b[~m] == 0.0) or np.array_equal(a[~m & (b != 0)], b[~m & (b != 0)])  # EDGE jobs still report all four edge sums
        assert O.score_from_sums(a, w, h) == O.score_from_sums(b, w, h)


@pytest.mark.parametrize("w,h", [(200, 180), (64, 40), (11, 11), (33, 12), (70, 300)])
def test_ssim_kernels_match_oracle(w, h):
    """SSIM / MS-SSIM stage (tm_ssim_kernels.h) on the CPU lane emulator: the u8 planes written by the ingest kernel are
    the oracle's quantised frames bit for bit; the per-scale sums agree to 1e-12; the finishing functions agree exactly."""
    frames = nv12_frames(w, h)
    for f in frames:
        f[0]["matrix"] = f[1]["matrix"] = 0
    r8, d8 = tm.synth.rgb8_pair(w, h)
    frames.append((dict(kind="rgb8", data=r8), dict(kind="rgb8", data=d8)))
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), ssim_window=O.ssim_window())
    for slot, (fr, fd) in enumerate(frames):
        lin = [oracle_linear(fr, w, h), oracle_linear(fd, w, h)]
        for side in range(2):
            q = O.quantize_u8(lin[side])
            for c in range(3):
                assert np.array_equal(em.qplane(slot, side, c), q[c]), ("q", slot, side, c)
        want = O.msssim_sums(lin[0], lin[1])
        np.testing.assert_allclose(em.ssim_sums(slot), want, rtol=1e-12, atol=1e-300)
        got_ssim = tm.engine.ssim_from_sums(em.ssim_sums(slot), w, h)
        assert got_ssim == O.ssim_from_sums(em.ssim_sums(slot), w, h)
        assert abs(got_ssim - O.ssim_from_sums(want, w, h)) <= 1e-7
        a, b = tm.engine.msssim_from_sums(em.ssim_sums(slot), w, h), O.msssim_from_sums(em.ssim_sums(slot), w, h)
        assert (a == b) or (np.isnan(a) and np.isnan(b))   # NaN below 176x176


@pytest.mark.parametrize("variant", [DEFAULT, WIDE_ROWS])
@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (64, 64), (1, 1), (2, 5), (129, 20), (65, 130), (200, 9), (257, 131)])
def test_default_pipeline_matches_oracle(w, h, variant):
    """k_ingest_wave + k_ingest_upper_rd (interleaved pyramid, no transposed copy) -> k_blur_v_jobs -> k_blur_h_jobs_x (the row pass
    transposes ref / dis itself), both row-pass instantiations, NV12 and RGB8 slots in one launch (per-frame dispatch)"""
    frames = nv12_frames(w, h)
    r8, d8 = tm.synth.rgb8_pair(w, h)
    frames.append((dict(kind="rgb8", data=r8), dict(kind="rgb8", data=d8)))
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=True,
                    ssim_window=O.ssim_window() if min(w, h) >= 11 else None)
    check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
    pr = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=False)
    m = weight_mask()
    for slot in range(len(frames)):
        assert np.array_equal(em.sums(slot)[m], pr.sums(slot)[m])
        if em.sg is not None:
            lin = [oracle_linear(frames[slot][0], w, h), oracle_linear(frames[slot][1], w, h)]
            for side in range(2):
                q = O.quantize_u8(lin[side])
                for c in range(3):
                    assert np.array_equal(em.qplane(slot, side, c), q[c])


def p016_frames(w, h, count=2):
    frames = []
    for n in range(count):
        (rs, rp, rch), (ds, dp, dch) = tm.synth.p016_pair(w, h, n + 1)
        frames.append((dict(kind="p016", data=rs, pitch=rp, coded_height=rch, matrix=(n + 1) % 3),
                       dict(kind="p016", data=ds, pitch=dp, coded_height=dch, matrix=n % 3)))  # ref and dis with different matrices
    return frames


@pytest.mark.parametrize("rows", [2, 4, 6, 8, 16])  # 4 and 8: pyramid levels 2..5 in the kernel's own epilogue (FOLD); the others through k_ingest_upper_rd
@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (1, 1), (2, 5), (129, 20), (257, 131), (130, 7)])
def test_row_walking_ingest_matches_oracle_and_the_tile_kernel(w, h, rows):
    """k_ingest_rows (what a launch of one 4:2:0 kind gets: a lane = one quad of BOTH frames, a wave = 64 quads x `rows` quad rows,
    level-2 pixels paired across iterations and across lane ^ 1): every XYB plane, the u8 planes, the integer SSE and the sums
    against the oracle; bit-identical arenas with k_ingest_wave (TM_VARIANT_TILE_INGEST) for NV12 and P016"""
    for frames in (nv12_frames(w, h), p016_frames(w, h)):
        ssimw = O.ssim_window() if min(w, h) >= 11 else None
        em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True, ssim_window=ssimw, ingest_rows=rows)
        check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
        old = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=TILE_INGEST, weights=O.weights(), full_sums=True, ssim_window=ssimw)
        assert np.array_equal(em.XYB.view(np.uint32), old.XYB.view(np.uint32)) and np.array_equal(em.SSE, old.SSE)
        assert np.array_equal(em.SUMS, old.SUMS)
        if ssimw is not None:
            assert np.array_equal(em.QU8, old.QU8)


UPPER_KERNEL = 0x2000


@pytest.mark.parametrize("rows", [4, 8])
@pytest.mark.parametrize("w,h", [(300, 261), (513, 130), (127, 65), (64, 64), (31, 33), (260, 17)])
def test_pyramid_levels_2_to_5_from_the_ingest_kernels_epilogue_equal_the_second_kernels(w, h, rows):
    """round 6: k_ingest_rows<.., FOLD> finishes levels 2..5 from an LDS tile of its workgroup's level-2 linear pixels --
    several workgroups in x and y, partial tiles on both edges, odd sizes at every level: the whole XYB arena, the SSE and the sums are the
    bits of the LIN2 + k_ingest_upper_rd arrangement (TM_VARIANT_UPPER_KERNEL), for 8-bit, 16-bit and the packed 10-bit kind"""
    for frames in (nv12_frames(w, h), p016_frames(w, h, 1), packed10_frames(w, h, 1)):
        new = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True, ingest_rows=rows)
        old = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=UPPER_KERNEL, weights=O.weights(), full_sums=True, ingest_rows=rows)
        assert np.array_equal(new.XYB.view(np.uint32), old.XYB.view(np.uint32)) and np.array_equal(new.SSE, old.SSE) and np.array_equal(new.SUMS, old.SUMS)
    if w * h < 40000:
        check_against_oracle(new, frames, w, h, have_linear=False, have_xybt=False)


@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (1, 1), (2, 5), (129, 20), (257, 131), (16, 200), (12, 64)])
def test_multi_wave_row_pass_writes_the_same_partial_sums(w, h):
    """k_blur_h_jobs_split (small launches: eight waves per 64-row block -- five producers of one recurrence each, a wave that fetches and
    transposes the ref / dis blocks, two consumers --, a 16-step LDS ring between them) against the oracle, and bit-identical 108 sums
    and PART entries with the one-wave row pass -- pruned and full job tables"""
    frames = nv12_frames(w, h)
    for full in (True, False):
        em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=SPLIT_ROWS, weights=O.weights(), full_sums=full)
        one = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=full)
        assert np.array_equal(em.SUMS, one.SUMS) and np.array_equal(em.PART, one.PART)
        if full:
            check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
        # the column pass of small launches: every role-wave a workgroup of its own -- the same pass-1 planes, bit for bit
        solo = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=SPLIT_ROWS | 0x10000, weights=O.weights(), full_sums=full)
        assert np.array_equal(solo.V.view(np.uint32), one.V.view(np.uint32)) and np.array_equal(solo.PART, one.PART)


FUSED_EDGE = 0x4000


@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (1, 1), (2, 5), (129, 20), (257, 131), (16, 200), (32, 32), (31, 96), (100, 33)])
def test_fused_edge_kernel_writes_the_same_sums_as_the_two_passes(w, h):
    """k_blur_edge_fused + k_finish_edge (both recurrences, the edge maps and their sums of an EDGE job in one kernel: bands of 32
    rows chained through {value, tag} words, tiles of 32 columns through LDS) against the two-pass kernels: the 108 sums bit for
    bit -- with the reference's weights (EDGE = scale 0 of X and B) and with a table that makes EVERY job an EDGE job (all six
    scales, bands of every height) -- and against the oracle"""
    frames = nv12_frames(w, h)
    only_edges = O.weights().reshape(3, 6, 6).copy()
    only_edges[:, :, 0] = 0.0; only_edges[:, :, 3] = 0.0   # no ssim weight anywhere ...
    only_edges[:, :, 1] = 1.0                             # ... and an edge weight everywhere
    for weights in (O.weights(), only_edges.ravel()):
        two = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=weights, full_sums=False)
        one = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=FUSED_EDGE, weights=weights, full_sums=False)
        assert np.array_equal(one.SUMS, two.SUMS)
        # what the engine launches: four adjacent bands of one plane per workgroup, the state through an LDS mailbox inside the group
        grp = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=FUSED_EDGE | 0x8000, weights=weights, full_sums=False)
        assert np.array_equal(grp.SUMS, two.SUMS)
    full = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
    mask = weight_mask()
    for slot in range(len(frames)):
        assert np.array_equal(one.sums(slot)[:, [1, 2, 4, 5], :], full.sums(slot)[:, [1, 2, 4, 5], :])  # every edge sum of every scale
    one = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=FUSED_EDGE, weights=O.weights(), full_sums=False)
    for slot in range(len(frames)):
        assert np.array_equal(one.sums(slot)[mask], full.sums(slot)[mask])


def planar_frames(w, h, bits, count=2):
    frames = []
    dt = np.uint8 if bits == 8 else np.uint16
    for n in range(count):
        ref, dis = tm.synth.yuv420_pair(w, h, n + 3, bits)
        frames.append(tuple(dict(kind="i420", data=tuple(p.astype(dt) for p in side), bits=bits, matrix=n % 3) for side in (ref, dis)))
    return frames


@pytest.mark.parametrize("variant", [DEFAULT, REFERENCE])
@pytest.mark.parametrize("bits", [8, 10])
@pytest.mark.parametrize("w,h", [(70, 38), (33, 67), (129, 20)])
def test_planar_i420_equals_the_repacked_biplanar_surface(w, h, bits, variant):
    """tm_engine_set_frame_i420's kinds: planar 4:2:0 (8-bit; 10-bit values in the low bits of u16) give exactly the planes and
    sums of the NV12 / P016 surface the same samples would be repacked into"""
    frames = planar_frames(w, h, bits)
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=variant == REFERENCE, have_xybt=variant == REFERENCE)


def packed10_frames(w, h, count=2, extra_words=0):
    frames = []
    for n in range(count):
        ref, dis = tm.synth.yuv420_pair(w, h, n + 5, 10)
        frames.append(tuple(dict(kind="i420p10", data=tuple(tm.synth.p10_pack_plane(p, tm.synth.p10_row_words(p.shape[1]) + extra_words) for p in side), matrix=(n + 1) % 3)
                            for side in (ref, dis)))
    return frames


@pytest.mark.parametrize("variant", [DEFAULT, TILE_INGEST, REFERENCE])
@pytest.mark.parametrize("w,h,extra", [(70, 38, 0), (33, 67, 2), (129, 20, 0), (400, 18, 0), (770, 12, 4)])  # one run, a partial block, beyond one block, chroma beyond a block
def test_packed_10_bit_planes_equal_the_planar_10_bit_frame(w, h, extra, variant):
    """tm_engine_set_frame_i420p10 (round 6: 10.7 instead of 16 bits per sample over PCIe): three samples per word, runs of 128 -- the
    row-walking kernel (a wave = one run behind one shift), the tile kernel and the reference kernel (per-lane word and shift) all give
    the planes and sums of the same samples handed over as planar 10-bit"""
    frames = packed10_frames(w, h, extra_words=extra)
    em = E.Emulated(w, h, frames, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=variant == REFERENCE, have_xybt=variant == REFERENCE)
    if variant == DEFAULT:  # ... and bit for bit what the 16-bit planar kind computes
        planar = []
        for fr in frames:
            cw = (w + 1) // 2
            planar.append(tuple(dict(kind="i420", bits=10, matrix=f["matrix"], data=tuple(tm.synth.p10_unpack_plane(p, n).astype(np.uint16) for p, n in zip(f["data"], (w, cw, cw)))) for f in fr))
        em2 = E.Emulated(w, h, planar, O.srgb8_lut(), coef_table(), variant=variant, weights=O.weights(), full_sums=True)
        assert np.array_equal(em.XYB, em2.XYB) and np.array_equal(em.SUMS, em2.SUMS) and np.array_equal(em.SSE, em2.SSE)


def test_planar_and_biplanar_frames_in_one_launch():
    """per-frame dispatch (mixed kinds): planar 8-bit, planar 10-bit with stray high bits masked away, NV12"""
    w, h = 46, 30
    frames = planar_frames(w, h, 8, 1) + planar_frames(w, h, 10, 1) + nv12_frames(w, h, 1) + packed10_frames(w, h, 1)
    dirty = tuple(dict(f, data=tuple((p | np.uint16(0xFC00)) for p in f["data"])) for f in frames[1])  # bits above the declared depth
    em = E.Emulated(w, h, frames[:1] + [dirty] + frames[2:], O.srgb8_lut(), coef_table(), variant=DEFAULT, weights=O.weights(), full_sums=True)
    check_against_oracle(em, frames, w, h, have_linear=False, have_xybt=False)
