"""The second restatement (oracle/twin_numpy.py, numpy, written from the reference's files) against the first
(oracle/tm_oracle.c, C): every f32 intermediate plane bit for bit, the f64 sums to 1e-12, the score to 1e-9 -- SURVEY.md 8c.
The closed libdevice functions' stand-ins (cbrt, transfer function) of the C oracle are plugged into the twin for the strict
bit-for-bit comparison; since round 3 they are correctly rounded but for a handful of arguments, so the twin with its OWN
float64-then-rounded functions -- sharing nothing at all with the C oracle -- gives the same planes too (last test)."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import oracle as O
from oracle import twin_numpy as T
from tm_pkg import tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def _round_f32_exact(q: Fraction) -> np.float32:
    """correctly rounded (nearest even) binary32 of a rational, by integer arithmetic only (normal range)"""
    if q == 0:
        return F(0.0)
    sign = -1 if q < 0 else 1
    q = abs(q)
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    scaled = q / Fraction(2) ** (e - 23)  # in [2^23, 2^24)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return F(sign * float(n) * 2.0 ** (e - 23))


def test_fma32_is_exactly_one_rounding():
    rng = np.random.default_rng(5)
    a = rng.standard_normal(4000).astype(F)
    b = rng.standard_normal(4000).astype(F)
    c = (-(a.astype(np.float64) * b.astype(np.float64)) * (1 + rng.standard_normal(4000) * 1e-4)).astype(F)  # heavy cancellation
    c[::3] = rng.standard_normal(len(c[::3])).astype(F)
    # constructed double-rounding traps: a*b + c whose float64 sum lands exactly on a binary32 boundary
    a2 = np.array([1.0 + 2.0 ** -23, 1.0 + 2.0 ** -23, 3.0, 1.0 - 2.0 ** -24], F)
    b2 = np.array([1.0 + 2.0 ** -23, 1.0 - 2.0 ** -23, 2.0 ** -25, 1.0 + 2.0 ** -23], F)
    c2 = np.array([2.0 ** -24 * (1 + 2.0 ** -23), 2.0 ** -24, 1.0 + 2.0 ** -24 * 0, 2.0 ** -24], F)
    a, b, c = np.concatenate([a, a2]), np.concatenate([b, b2]), np.concatenate([c, c2])
    got = T.fma32(a, b, c)
    for i in range(len(a)):
        want = _round_f32_exact(Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i])))
        assert got[i] == want, (i, a[i], b[i], c[i], got[i], want)


def test_constants_derived_by_the_twin_equal_the_references_literals():
    tab = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["gaussian"]
    radius, mul_in, mul_prev, mul_prev2 = T.gaussian_constants()
    assert radius == tab["RADIUS"]
    for i, k in enumerate((1, 3, 5)):
        assert int(mul_in[i].view(np.uint32)) == tab[f"MUL_IN_{k}"]
        assert int(mul_prev[i].view(np.uint32)) == tab[f"MUL_PREV_{k}"]
        assert int(mul_prev2[i].view(np.uint32)) == tab[f"MUL_PREV2_{k}"]
    for m in range(3):
        assert T.kr_kb(m) == O.kr_kb(m)
        for bits in (8, 16):
            assert np.array_equal(np.array(T.yuv_coefficients(m, bits), F), O.yuv_coefficients(m, bits))


def _pair(w, h, n):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    return O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 8, 0), O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 8, 0)


@pytest.mark.parametrize("w,h", [(64, 48), (67, 35)])
def test_every_intermediate_plane_of_the_two_restatements_is_bit_identical(w, h):
    lr, ld = _pair(w, h, 3)
    t_sums, cap = T.ssimulacra2_sums(lr, ld, cbrt=O.cbrtf, capture=True)
    o_sums, pyr = O.ssimulacra2_sums(lr, ld, want_xyb=True)
    for s in range(6):
        for side in range(2):
            assert np.array_equal(cap[s]["xyb"][side], pyr[s][side]), ("xyb", s, side)
        _, oc = O.process_scale(pyr[s][0], pyr[s][1], capture=True)
        for key, n in (("pass1", 5), ("pass2", 5), ("maps", 3)):
            for i in range(n):
                assert np.array_equal(np.asarray(cap[s][key][i]), oc[key][i]), (key, s, i)
    np.testing.assert_allclose(t_sums, o_sums, rtol=1e-12, atol=1e-300)
    assert abs(T.score_from_sums(t_sums, w, h) - O.score_from_sums(o_sums, w, h)) <= 1e-9
    assert T.score_from_sums(o_sums, w, h) == O.score_from_sums(o_sums, w, h)  # post-processing alone: same f64 operations


def test_stage_functions_agree_one_by_one():
    rng = np.random.default_rng(11)
    lin = rng.random((3, 37, 53)).astype(F)
    assert np.array_equal(T.linear_to_xyb(lin, cbrt=O.cbrtf), O.linear_to_xyb(lin))
    assert np.array_equal(T.downscale_by_2(lin), np.stack([O.downscale_by_2(lin[c]) for c in range(3)]))
    assert np.array_equal(T.blur_pass(lin[0]), O.blur_columns(lin[0]))
    planes = [rng.random((29, 31)).astype(F) * F(0.5) for _ in range(7)]
    for a, b in zip(T.error_maps(*planes), O.error_maps(*planes)):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind,matrix", [("nv12", 0), ("nv12", 1), ("nv12", 2), ("p016", 0), ("p016", 2)])
def test_yuv_to_linear_is_bit_identical_with_the_shared_transfer_function(kind, matrix):
    w, h = 70, 38
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    (rs, rp, rch), _ = gen(w, h, 5)
    bits = 8 if kind == "nv12" else 16
    want = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, matrix)
    got = T.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, matrix, eotf=O.bt709_eotf)
    assert np.array_equal(got, want)
    # with the twin's own transfer function -- the reference's expression evaluated as written, f32-rounded base (v + a) / A,
    # then a float64 pow rounded once -- the planes are THE SAME BITS (but for one sample in ~130 000 whose float64 cubic falls on
    # the other side of a rounding boundary): the stand-in is that expression, correctly rounded (round 2, a fit in v: up to 5 ulp)
    exact = T.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, matrix, eotf="exact")
    ulp = np.abs(exact.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp > 0).sum() <= 2


def test_the_twin_with_its_own_exact_functions_reproduces_the_oracle_end_to_end():
    """nothing shared: YUV -> linear with the twin's float64 pow of the f32 base, XYB with the twin's float64 cbrt -- against the C
    oracle's binary64 cubic and 21-operation f32 cube root: the same planes (a sample in ~1e5 may sit on the other side of a rounding
    boundary) and the same score to 1e-4 (four of the seven frozen 1080p / 640x360 goldens agree to the last bit)"""
    w, h = 96, 64
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 9)
    lin_t = [T.yuv420_biplanar_to_linear(s, p, c, w, h, 8, 0, eotf="exact") for s, p, c in ((rs, rp, rch), (ds, dp, dch))]
    lin_o = [O.yuv420_biplanar_to_linear(s, p, c, w, h, 8, 0) for s, p, c in ((rs, rp, rch), (ds, dp, dch))]
    for a, b in zip(lin_t, lin_o):
        assert (a.view(np.uint32) != b.view(np.uint32)).sum() <= 1
    t_sums, cap = T.ssimulacra2_sums(lin_t[0], lin_t[1], cbrt="exact", capture=True)
    o_sums, pyr = O.ssimulacra2_sums(lin_o[0], lin_o[1], want_xyb=True)
    differing = sum(int((np.asarray(cap[s]["xyb"][side]).view(np.uint32) != pyr[s][side].view(np.uint32)).sum()) for s in range(6) for side in range(2))
    assert differing <= 8  # of 2 x 3 x 8 190 samples
    assert abs(T.score_from_sums(t_sums, w, h) - O.score_from_sums(o_sums, w, h)) <= 1e-4
