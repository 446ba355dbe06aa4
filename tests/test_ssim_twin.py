"""SSIM / MS-SSIM have two witnesses: oracle/tm_ssim.c (f32 per pixel, operation by operation, what the HIP kernels are
held to bit for bit) against oracle/twin_ssim.py (float64 scipy, written from Wang 2004 / Wang 2003 without reading the C
file).  A shared misreading of the papers -- window normalisation, the `valid` extent, C1/C2, the decimation phase, the
exponent table, which term a scale contributes -- would have to be made twice, independently, to stay green here.

Tolerances: the C statement rounds every per-pixel quantity to f32 (sigma = E[x^2] - mu^2 cancels on 16-bit magnitudes),
the twin is float64 throughout: per-scale window means agree to 5e-6 relative (observed <= 1.7e-6, the worst being the
single-window scale of a 176 x 176 picture), scores to 1e-6 absolute (observed <= 2.3e-7); a single window over a flat
patch is the f32 statement's worst case, 1.3e-5 (test_hand_computed_single_window).  The one choice the papers leave
open -- an odd row at a decimation step -- moves a mean by >= 1.7e-5 (test_decimation_choice_is_visible).
"""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import twin_ssim as T


def synth_pair(w, h, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([0.5 + 0.3 * np.sin(xx / 17.0 + c) * np.cos(yy / 23.0 - c) for c in range(3)]).astype(np.float32)
    ref = np.clip(base + r.normal(0, 0.05, base.shape), 0, 1).astype(np.float32)
    dis = np.clip(np.round(ref * 32) / 32 + r.normal(0, 0.02, base.shape), 0, 1).astype(np.float32)
    return ref, dis


def rgb8_pair(w, h, seed):
    r = np.random.default_rng(seed)
    ref8 = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref8[: h // 2] = (ref8[: h // 2] // 4) + 96  # a flat-ish half so that the luminance term matters
    dis8 = np.clip(ref8.astype(np.int32) + r.integers(-9, 10, ref8.shape), 0, 255).astype(np.uint8)
    return O.rgb8_to_linear(ref8), O.rgb8_to_linear(dis8)  # (3, h, w) linear f32 through the reference's 256-entry table


def window_counts(w, h):
    out = []
    for _ in range(5):
        out.append(max(w - 10, 0) * max(h - 10, 0))
        w, h = w // 2, h // 2
    return np.asarray(out, np.float64)


CASES = [(176, 176, "synth"), (177, 233, "synth"), (333, 203, "synth"), (640, 360, "synth"), (1920, 1080, "synth"),
         (208, 190, "rgb8")]


@pytest.mark.parametrize("w,h,kind", CASES)
def test_c_statement_agrees_with_the_float64_twin(w, h, kind):
    if kind == "rgb8":
        ref, dis = rgb8_pair(w, h, 7)
    else:
        ref, dis = synth_pair(w, h, w * 31 + h)
    ssim_c, ms_c, sums = O.ssim_msssim(ref, dis)
    means_c = sums / window_counts(w, h)[None, :, None]
    means_t = T.scale_means(ref, dis, 5, odd="drop")
    assert np.all(np.isfinite(means_t))
    np.testing.assert_allclose(means_c, means_t, rtol=5e-6, atol=0)
    assert abs(ssim_c - T.ssim(ref, dis)) <= 1e-6
    assert abs(ms_c - T.msssim(ref, dis)) <= 1e-6
    assert 0.0 < ms_c < 1.0 and 0.0 < ssim_c < 1.0


def test_quantisation_is_round_half_even_on_both_sides():
    v = (np.arange(0, 511, dtype=np.float32) / np.float32(510.0)).reshape(1, 1, -1).repeat(3, axis=0)  # k/2 * 1/255 in [0, 1]: every tie
    q_c = O.quantize_u8(np.ascontiguousarray(v))
    np.testing.assert_array_equal(q_c.astype(np.float64), T.quantize(v))


def test_identical_frames_score_exactly_one():
    ref, _ = synth_pair(192, 176, 5)
    s, m, sums = O.ssim_msssim(ref, ref)
    assert s == 1.0 and m == 1.0
    assert T.ssim(ref, ref) == 1.0 and T.msssim(ref, ref) == 1.0
    np.testing.assert_array_equal(sums[..., 0], sums[..., 1])


def test_hand_computed_single_window():
    """11 x 11: one window position.  Constant planes a, b: sigma = 0, cs = 1, ssim = (2ab + C1) / (a^2 + b^2 + C1).
    Arbitrary x and y = x + d: sigma_xy = sigma_x^2 = sigma_y^2, cs = 1, ssim = the luminance term of (mu, mu + d) with
    mu = sum(window * x) -- evaluated here from the closed form of the window, neither statement's code."""
    c1 = (0.01 * 255) ** 2
    const = lambda v: np.full((3, 11, 11), v / 255.0, np.float32)
    s, _, sums = O.ssim_msssim(const(100), const(50))
    want = (2 * 100 * 50 + c1) / (100 ** 2 + 50 ** 2 + c1)  # 0.800103...
    assert abs(want - 0.8001039) < 1e-7
    # flat planes are the f32 statement's worst case: sigma = E[x^2 + y^2] - mu_x^2 - mu_y^2 cancels 12 500 against 12 500 and what
    # is left (~1e-3) stands beside C2 = 58.5 -> 1.3e-5 on this single window; the float64 twin has the closed form to 1e-12
    assert abs(s - want) <= 3e-5 and abs(T.ssim(const(100), const(50)) - want) <= 1e-12
    np.testing.assert_allclose(sums[:, 0, 1], 1.0, rtol=3e-5)
    assert np.all(sums[:, 1:] == 0.0)

    r = np.random.default_rng(3)
    x8 = r.integers(20, 200, (11, 11)).astype(np.float64)
    k = np.arange(11) - 5.0
    g = np.exp(-k * k / 4.5)
    g /= g.sum()  # separable form of the same normalised window
    mu = float(g @ x8 @ g)
    d = 23.0
    want = (2 * mu * (mu + d) + c1) / (mu * mu + (mu + d) ** 2 + c1)
    x = np.repeat((x8 / 255.0).astype(np.float32)[None], 3, axis=0)
    y = np.repeat(((x8 + d) / 255.0).astype(np.float32)[None], 3, axis=0)
    s, _, sums = O.ssim_msssim(x, y)
    assert abs(s - want) <= 3e-5 and abs(T.ssim(x, y) - want) <= 1e-12
    np.testing.assert_allclose(sums[:, 0, 1], 1.0, rtol=3e-5)


def test_window_is_the_normalised_outer_product():
    g = O.ssim_window().astype(np.float64)
    assert abs(g.sum() - 1.0) < 1e-7
    np.testing.assert_allclose(np.outer(g, g), T.window(), rtol=2e-7)
    assert np.argmax(g) == 5 and np.all(g == g[::-1])
    assert abs(g[5] / g[4] - np.exp(1 / 4.5)) < 1e-6  # sigma = 1.5


def test_decimation_choice_is_visible():
    """The papers do not say what an odd row does at a 2x decimation.  The build drops it (floor); replicating it (ceil, the
    authors' imfilter 'symmetric' + 1:2:end) moves the scale-2 means by 2e-5 .. 4e-5: small, and still outside the tolerance above."""
    ref, dis = synth_pair(177, 233, 11)
    _, _, sums = O.ssim_msssim(ref, dis)
    means_c = sums / window_counts(177, 233)[None, :, None]
    alt = T.scale_means(ref, dis, 5, odd="clamp")
    np.testing.assert_allclose(means_c[:, 0], alt[:, 0], rtol=5e-6)
    assert np.abs(means_c[:, 1, 1] / alt[:, 1, 1] - 1).min() > 1e-5


def test_minimum_sizes():
    ref, dis = synth_pair(175, 200, 1)  # 175 >> 4 = 10 < 11: no fifth scale
    s, m, _ = O.ssim_msssim(ref, dis)
    assert np.isnan(m) and 0 < s < 1
    assert np.isnan(T.scale_means(ref, dis, 5)[:, 4]).all()
    assert abs(s - T.ssim(ref, dis)) <= 1e-6
