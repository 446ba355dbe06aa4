"""Pin the CPU oracle on every known-answer datum the reference holds for this path
(SURVEY.md 8c): sRGB LUT, recursive-Gaussian literals, the 108 weights, the NPP `sum_value`
test, identical-input behaviour, plus accuracy of the restated libdevice functions."""
import os
import subprocess
import sys

import numpy as np

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_srgb_lut_matches_reference_table(golden_tables):
    lut = O.srgb8_lut()
    assert (lut.view(np.uint32) == np.array(golden_tables["srgb_lut_bits"], np.uint32)).all()
    assert lut[0] == 0.0 and lut[255] == 1.0 and np.all(np.diff(lut) > 0)


def test_srgb_lut_close_to_formula():
    # the table is the gamma-2.4 sRGB EOTF evaluated at i/255 (reference srgb.rs:40-48)
    lut = O.srgb8_lut().astype(np.float64)
    i = np.arange(256) / 255.0
    f = np.where(i <= 0.04045, i / 12.92, ((i + 0.055) / 1.055) ** 2.4)
    assert np.abs(lut - f).max() < 5e-7


def test_weights_match_reference_table(golden_tables):
    w = O.weights()
    ref = np.array([float(x) for x in golden_tables["weights"]])
    assert (w == ref).all()
    assert int((w != 0).sum()) == 52


def test_gaussian_literals_match_reference_and_derivation(golden_tables):
    g = golden_tables["gaussian"]
    k = O.gaussian_constants().view(np.uint32)
    names = ["MUL_IN_1", "MUL_IN_3", "MUL_IN_5", "MUL_PREV_1", "MUL_PREV_3", "MUL_PREV_5", "MUL_PREV2_1"]
    assert [int(v) for v in k] == [g[n] for n in names]
    assert g["RADIUS"] == 5
    # independent derivation from the paper's equations (tools/derive_gaussian.py)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import derive_gaussian as D
    radius, n2, d1 = D.derive()
    assert radius == 5
    assert [D.f32_bits(v) for v in n2] == [g["MUL_IN_1"], g["MUL_IN_3"], g["MUL_IN_5"]]
    assert [D.f32_bits(-v) for v in d1] == [g["MUL_PREV_1"], g["MUL_PREV_3"], g["MUL_PREV_5"]]
    assert [D.f32_bits(v) for v in d1] == [g["VERT_MUL_PREV_1"], g["VERT_MUL_PREV_3"], g["VERT_MUL_PREV_5"]]


def test_blur_is_normalised_and_finite_support():
    # impulse response of one pass = truncated-cosine Gaussian: sums to ~1, support radius 5
    h = 64
    p = np.zeros((h, 1), np.float32)
    p[30, 0] = 1.0
    o = O.blur_columns(p)[:, 0].astype(np.float64)
    assert abs(o.sum() - 1.0) < 1e-5
    assert np.abs(o[:24]).max() < 1e-6 and np.abs(o[37:]).max() < 1e-5
    assert abs(o[30] - 0.2659) < 5e-3  # ~ 1/(sigma sqrt(2 pi)) with sigma 1.5
    assert np.allclose(o[25:36], o[25:36][::-1], atol=1e-6)


def test_npp_sum_known_answer(golden_tables):
    # cudarse-npp/src/image/ist.rs:204-217: a 128x128 image set to (1,0,0) sums to [16384,0,0];
    # here: maps that are exactly (1,0,0) give those 1-norm sums.
    ones = np.ones((128, 128), np.float32)
    assert float(ones.astype(np.float64).sum()) == golden_tables["known_answers"]["npp_sum_128x128_r1"][0]


def test_identical_inputs():
    # With the GPU arithmetic the SSIM map is exactly 0 for identical inputs, the edge maps hold
    # fma rounding residue (d1 = fma(v, rn(1/v), -1) != 0), so the score is just below 100 --
    # consistent with the reference README's "max 99.99" for its sample run.
    rng = np.random.default_rng(3)
    a = rng.random((3, 40, 56), dtype=np.float32)
    score, sums = O.ssimulacra2_from_linear(a, a)
    assert np.all(sums[:, 0, :] == 0.0) and np.all(sums[:, 3, :] == 0.0)
    assert 99.9 < score <= 100.0


def test_score_polynomial_known_points():
    z = np.zeros(108)
    assert O.score_from_sums(z, 64, 64) == 100.0


def test_cbrtf_is_the_correctly_rounded_cube_root_but_for_a_handful_of_arguments():
    """21-operation f32 cube root (oracle/tm_math.h == tm_device_math.h cbrt_core2): every float of [1, 8) -- all mantissas for
    each exponent residue mod 3 -- against long-double cbrtl: 11 of 25 M are not the nearest float (round 2's 20 operations: 933)"""
    worst, bad = O.cbrtf_scan(1.0, 8.0)
    assert worst < 0.50001 and bad <= 11, (worst, bad)
    from oracle import twin_numpy as T
    x = np.random.default_rng(5).uniform(0.0037, 1.004, 500000).astype(np.float32)
    assert (O.cbrtf(x) != T.cbrt_exact(x)).sum() <= 2  # the twin's float64 cbrt, rounded once: the same bits
    a = np.concatenate([10 ** np.random.default_rng(1).uniform(-44, 38, 4000), [0.0, 1.0, 8.0, 27.0, 0.0037, 1.004]]).astype(np.float32)
    got = O.cbrtf(a)
    want = np.cbrt(a.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    assert (np.abs(got.astype(np.float64) - want) <= 0.5003 * ulp + 1e-300).all()  # scaled / special arguments
    assert got[4000] == 0.0 and got[4001] == 1.0 and got[4002] == 2.0 and got[4003] == 3.0


def test_powf_is_correctly_rounded_on_samples():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(0.05, 1.3, 20000), 10 ** rng.uniform(-3, 3, 2000)]).astype(np.float32)
    for y in (np.float32(1.0) / np.float32(0.45), np.float32(2.4)):
        got = O.powf(x, y)
        want = np.power(x.astype(np.float64), float(y)).astype(np.float32)
        bad = got != want
        # double-rounding ties are the only allowed deviation (none observed)
        assert bad.sum() == 0


def test_bt709_transfer_function_is_the_correctly_rounded_reference_expression():
    """the power branch: the reference's f32 base x = (v + a) / A (its two f32 operations), then x^(1/0.45f) from 428 binary64 cubics,
    rounded once (oracle/tm_math.h == tm_device_math.h): scanned over EVERY float v of [threshold, 1) against long-double powl of
    that f32 base -- the reference's expression as written, with an exact pow"""
    worst, at, off = O.bt709_eotf_max_ulp()
    assert worst < 0.5001, (worst, at)
    assert off <= 150  # of 15.4 M arguments: 117 are not the correctly rounded value (by < 0.0001 ulp)
    v = np.array([0.0, 0.01, 0.0812, 0.08124286, 0.5, 0.99999994, 1.0, 1.3], np.float32)
    got = O.bt709_eotf(v)
    assert got[0] == 0.0 and got[1] == np.float32(0.01) / np.float32(4.5) and got[2] == np.float32(0.0812) / np.float32(4.5)  # linear branch: IEEE division
    assert got[6] == 1.0 and got[7] == 1.0 and got[5] <= 1.0  # base >= 1: exactly 1 (the callers clamp)
    x = np.linspace(0.0813, 0.9999, 20001).astype(np.float32)
    y = O.bt709_eotf(x)
    assert (np.diff(y.astype(np.float64)) >= 0).all()  # monotone across the segment joints
    # the twin's own evaluation (float64 pow of the f32 base, rounded once) gives the same bits
    from oracle import twin_numpy as T
    rng = np.random.default_rng(3)
    v = rng.uniform(0.0813, 1.0, 400000).astype(np.float32)
    d = np.abs(O.bt709_eotf(v).view(np.int32).astype(np.int64) - T.bt709_eotf(v, "exact").view(np.int32).astype(np.int64))
    assert d.max() <= 1 and (d > 0).sum() <= 20


def test_yuv_matrix_constants():
    # Kr,Kb derived from chromaticities in f32
    exp = {0: (0.212639, 0.072192), 1: (0.212376, 0.086564), 2: (0.222004, 0.071341)}
    for m, (kr, kb) in exp.items():
        a, b = O.kr_kb(m)
        assert abs(float(a) - kr) < 2e-6 and abs(float(b) - kb) < 2e-6


def test_oracle_header_declares_role():
    src = open(os.path.join(ROOT, "oracle", "tm_oracle.c")).read()
    assert "TEST INFRASTRUCTURE ONLY" in src and "parity unpinned" in src


def test_cpu_path_restatement_properties():
    # examples/cpu.rs restated (oracle/tm_cpu_path.c): identical inputs -> exactly 100 (f64 maps, unlike the GPU
    # arithmetic), and it agrees with the GPU-arithmetic oracle within the reference's own 0.25 band
    # (examples/compare.rs:72) on an image large enough for all six scales
    from tm_pkg import tm
    r8, d8 = tm.synth.rgb8_pair(320, 256)
    assert O.cpu_path_score_srgb8(r8, r8) == 100.0
    c = O.cpu_path_score_srgb8(r8, d8)
    g, _ = O.ssimulacra2_from_linear(O.rgb8_to_linear(r8), O.rgb8_to_linear(d8))
    assert 0.0 < c < 100.0 and abs(c - g) < 0.25


def test_threaded_cpu_path_runner_reproduces_the_single_call():
    """tmo_cpu_path_run (bench.py's all-core CPU baseline: pthreads, one pair per worker at a time, workspace allocated once per
    worker) returns, for every pair, exactly the score of tmo_cpu_path_score_linear on the oracle's YUV -> linear conversion"""
    import numpy as np
    from oracle import oracle as O
    from tm_pkg import tm
    w, h = 160, 96
    pairs = [tm.synth.nv12_pair(w, h, n) for n in range(3)]
    want = []
    for (rs, rp, rch), (ds, dp, dch) in pairs:
        want.append(O.cpu_path_score_linear(O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 8, 0), O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 8, 0)))
    for threads in (1, 3, 7):
        secs, scores = O.cpu_path_run(pairs, w, h, 8, 11, threads)
        assert secs > 0 and np.array_equal(scores, np.resize(np.array(want), 11)), threads
    p16 = [tm.synth.p016_pair(w, h, 0)]
    secs, scores = O.cpu_path_run(p16, w, h, 16, 2, 2)
    (rs, rp, rch), (ds, dp, dch) = p16[0]
    assert scores[0] == scores[1] == O.cpu_path_score_linear(O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 16, 0), O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 16, 0))
