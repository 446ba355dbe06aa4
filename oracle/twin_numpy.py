"""oracle/twin_numpy.py -- TEST INFRASTRUCTURE ONLY: a SECOND, independently written restatement of the reference's
GPU arithmetic for the SSIMULACRA2 path, in numpy ("two independently written restatements agreeing
bit-for-bit on f32 intermediates").

Written from the reference's files, not from oracle/tm_oracle.c (paths relative to /root/reference/crates):
  colour      cuda-colorspace-kernel/src/biplanar.rs:8-70, lib.rs:97-132,186-236, constants.rs:3-18, const_algebra.rs:19-102
  xyb         ssimulacra2-cuda-kernel/src/xyb.rs:3-79
  downscale   ssimulacra2-cuda-kernel/src/downscale.rs:5-35
  blur        ssimulacra2-cuda-kernel/src/blur.rs:34-137, constants from build.rs:28-145
  error maps  ssimulacra2-cuda-kernel/src/error_maps.rs:5-60
  dataflow    ssimulacra2-cuda/src/lib.rs:140-229 (pyramid), 293-415 (process_scale), 417-447 (reduce), 586-622 (score)
The 108 weights are DATA of the reference (lib.rs:454-584) and are read from tests/golden/reference_tables.json.

All image arithmetic is IEEE binary32: numpy float32 element-wise + - * / are correctly rounded, and the reference's
`mul_add` (one rounding) is emulated EXACTLY by fma32() below.  The two closed NVIDIA libdevice functions the reference calls
are parameters of the twin:
  cbrt   "exact"  correctly rounded cube root (float64 cbrt rounded to float32) -- libdevice documents 1 ulp
         or a callable float32 array -> float32 array (the tests plug in the C oracle's 20-operation sequence to compare planes
         bit for bit; that sequence is itself pinned to <= 0.5003 ulp of the true cube root by tests/test_oracle_pins.py)
  eotf   "exact"      pow evaluated in float64, rounded once to float32
         "fast_powf"  the SHAPE of libdevice's __nv_fast_powf: exp2f(y * log2f(x)) with every step rounded to float32
         or a callable float32 array -> float32 array
The product never imports this module.
"""
import json
import math
import os

import numpy as np

F = np.float32
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALES = 6


# ---------------------------------------------------------------------------------------------------------------------
# exact binary32 fused multiply-add on arrays
# ---------------------------------------------------------------------------------------------------------------------
def fma32(a, b, c):
    """round_to_f32(a * b + c) with ONE rounding.  a * b is exact in float64 (24 + 24 significand bits); the float64 sum
    p + c is rounded once to 53 bits, and rounding that again to 24 bits can only differ from the single rounding when the
    float64 sum sits exactly on a float32 rounding boundary while the exact sum does not: TwoSum gives the exact residual,
    and such a sum is nudged off the boundary towards the exact value before the final conversion."""
    a = np.asarray(a, F).astype(np.float64)
    b = np.asarray(b, F).astype(np.float64)
    c = np.asarray(c, F).astype(np.float64)
    p = a * b
    s = np.atleast_1d(p + c)
    p, c = np.broadcast_arrays(np.atleast_1d(p), np.atleast_1d(c))
    bb = s - p
    err = (p - (s - bb)) + (c - bb)  # exact: s + err == p + c
    low = np.ascontiguousarray(s).view(np.uint64) & np.uint64(0x1FFFFFFF)
    on_boundary = (low == np.uint64(0x10000000)) & (err != 0.0)
    if on_boundary.any():
        s = np.where(on_boundary, np.nextafter(s, np.where(err > 0.0, np.inf, -np.inf)), s)
    out = s.astype(F)
    tiny = (out != 0) & (np.abs(out) < np.finfo(F).tiny)
    assert not tiny.any(), "subnormal binary32 result: the boundary test above assumes normal numbers"
    return out


# ---------------------------------------------------------------------------------------------------------------------
# recursive Gaussian constants, build.rs:28-145 (Charalampidis 2016, sigma 1.5); float64 like the build script
# ---------------------------------------------------------------------------------------------------------------------
def gaussian_constants():
    """-> (radius, MUL_IN[3], MUL_PREV[3], MUL_PREV2[3]) as float32, for k = 1, 3, 5"""
    sigma = 1.5
    radius = float(round(3.2795 * sigma + 0.2546))  # (57)
    pi_div_2r = math.pi / (2.0 * radius)
    omega = [pi_div_2r, 3.0 * pi_div_2r, 5.0 * pi_div_2r]
    p = [1.0 / math.tan(0.5 * omega[0]), -1.0 / math.tan(0.5 * omega[1]), 1.0 / math.tan(0.5 * omega[2])]  # (37)
    r = [p[0] * p[0] / math.sin(omega[0]), -p[1] * p[1] / math.sin(omega[1]), p[2] * p[2] / math.sin(omega[2])]  # (44)
    rho = [math.exp(-0.5 * sigma * sigma * w * w) / radius for w in omega]  # (50)
    d13 = p[0] * r[1] - r[0] * p[1]  # (52)
    d35 = p[1] * r[2] - r[1] * p[2]
    d51 = p[2] * r[0] - r[2] * p[0]
    zeta15, zeta35 = d35 / d13, d51 / d13
    A = np.array([[p[0], p[1], p[2]], [r[0], r[1], r[2]], [zeta15, zeta35, 1.0]], np.float64)  # (56)
    gamma = np.array([1.0, radius * radius - sigma * sigma, zeta15 * rho[0] + zeta35 * rho[1] + rho[2]], np.float64)  # (55)
    beta = np.linalg.solve(A, gamma)  # (53)
    assert abs(beta[0] * p[0] + beta[1] * p[1] + beta[2] * p[2] - 1.0) < 1e-12  # (39), the build script's own check
    n2 = [-beta[i] * math.cos(omega[i] * (radius + 1.0)) for i in range(3)]  # (33)
    d1 = [-2.0 * math.cos(omega[i]) for i in range(3)]
    mul_in = np.array(n2, F)
    mul_prev = np.array([-d for d in d1], F)
    mul_prev2 = np.array([-1.0, -1.0, -1.0], F)
    return int(radius), mul_in, mul_prev, mul_prev2


_RADIUS, _MUL_IN, _MUL_PREV, _MUL_PREV2 = gaussian_constants()


def blur_pass(src):
    """blur_plane_pass_fused (blur.rs:34-137) on a (rows, columns) array: every COLUMN is one thread of the reference; the
    loop over y is the thread's loop, vectorised over the columns."""
    src = np.ascontiguousarray(src, F)
    h, n = src.shape
    N = _RADIUS
    ring_size = 2 * N + 1
    ring = np.zeros((ring_size, n), F)
    zero = np.zeros(n, F)
    prev = [zero.copy() for _ in range(3)]
    prev2 = [zero.copy() for _ in range(3)]
    out = np.zeros_like(src)
    for y in range(-N + 1, h):
        right = y + N - 1
        right_val = src[right] if right < h else zero
        left = y - N - 1
        left_val = ring[left % ring_size] if left >= 0 else zero
        s = left_val + right_val
        o = [None] * 3
        for k in range(3):
            v = s * _MUL_IN[k]
            v = fma32(_MUL_PREV2[k], prev2[k], v)
            prev2[k] = prev[k]
            v = fma32(_MUL_PREV[k], prev[k], v)
            prev[k] = v
            o[k] = v
        if y >= 0:
            out[y] = (o[0] + o[1]) + o[2]
        ring[right % ring_size] = right_val
    return out


# ---------------------------------------------------------------------------------------------------------------------
# XYB, xyb.rs:3-79
# ---------------------------------------------------------------------------------------------------------------------
_K_M02 = F(0.078)
_K_M00 = F(0.30)
_K_M01 = F(F(F(1.0) - _K_M02) - _K_M00)
_K_M12 = F(0.078)
_K_M10 = F(0.23)
_K_M11 = F(F(F(1.0) - _K_M12) - _K_M10)
_K_M20 = F(0.24342269)
_K_M21 = F(0.20476745)
_K_M22 = F(F(F(1.0) - _K_M20) - _K_M21)
_K_B0 = F(0.0037930734)
_K_B0_ROOT = F(0.1559542025327239180319220163705)
_OPSIN = [[_K_M00, _K_M01, _K_M02], [_K_M10, _K_M11, _K_M12], [_K_M20, _K_M21, _K_M22]]


def cbrt_exact(a):
    return np.cbrt(np.asarray(a, F).astype(np.float64)).astype(F)


def linear_to_xyb(lin, cbrt="exact"):
    """lin: (3, h, w) linear RGB -> (3, h, w) "positive XYB" (px_linear_rgb_to_positive_xyb)"""
    cb = cbrt_exact if cbrt == "exact" else cbrt
    r, g, b = (np.ascontiguousarray(lin[i], F) for i in range(3))
    mixed = [fma32(m[0], r, fma32(m[1], g, fma32(m[2], b, _K_B0))).reshape(r.shape) for m in _OPSIN]
    t = [np.asarray(cb(np.maximum(m, F(0.0))), F) - _K_B0_ROOT for m in mixed]
    x = F(0.5) * (t[0] - t[1])
    y = F(0.5) * (t[0] + t[1])
    return np.stack([fma32(x, F(14.0), F(0.42)).reshape(x.shape), y + F(0.01), (t[2] - y) + F(0.55)])


# ---------------------------------------------------------------------------------------------------------------------
# downscale_by_2, downscale.rs:5-35
# ---------------------------------------------------------------------------------------------------------------------
def downscale_by_2(img):
    """(3, h, w) -> (3, ceil(h/2), ceil(w/2)); sum starts at 0.0 and adds (iy, ix) = (0,0), (0,1), (1,0), (1,1)"""
    img = np.ascontiguousarray(img, F)
    _, h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    oy, ox = np.arange(dh), np.arange(dw)
    s = np.zeros((3, dh, dw), F)
    for iy in range(2):
        for ix in range(2):
            ys = np.minimum(oy * 2 + iy, h - 1)
            xs = np.minimum(ox * 2 + ix, w - 1)
            s = s + img[:, ys][:, :, xs]
    return s * F(0.25)


# ---------------------------------------------------------------------------------------------------------------------
# compute_error_maps, error_maps.rs:5-60
# ---------------------------------------------------------------------------------------------------------------------
def error_maps(source, distorted, mu1, mu2, sigma11, sigma22, sigma12):
    C2 = F(0.0009)
    one = F(1.0)
    mu11 = mu1 * mu1
    mu22 = mu2 * mu2
    mu12 = mu1 * mu2
    mu_diff = mu1 - mu2
    num_m = fma32(mu_diff, -mu_diff, one).reshape(mu1.shape)
    num_s = fma32(F(2.0), sigma12 - mu12, C2).reshape(mu1.shape)
    denom_s = ((sigma11 - mu11) + (sigma22 - mu22)) + C2
    ssim = np.maximum(one - (num_m * num_s) / denom_s, F(0.0))
    denom = one / (one + np.abs(source - mu1))
    numer = one + np.abs(distorted - mu2)
    d1 = fma32(numer, denom, F(-1.0)).reshape(mu1.shape)
    return ssim, np.maximum(d1, F(0.0)), np.maximum(-d1, F(0.0))


# ---------------------------------------------------------------------------------------------------------------------
# process_scale + reduce, ssimulacra2-cuda/src/lib.rs:293-447
# ---------------------------------------------------------------------------------------------------------------------
def _blur_all(planes):
    """one blur_pass_fused launch = the five images side by side (every column of every image is its own thread)"""
    n = planes[0].shape[2]
    wide = blur_pass(np.concatenate([p[c] for p in planes for c in range(3)], axis=1))
    return [np.stack([wide[:, (3 * i + c) * n:(3 * i + c + 1) * n] for c in range(3)]) for i in range(len(planes))]


def process_scale(ref_xyb, dis_xyb):
    """-> (sums [kind 6][channel 3] float64, dict of intermediate planes in the TRANSPOSED orientation the reference
    evaluates them in: pass1[5], pass2[5], maps[3], each (3, w, h))"""
    ref_xyb = np.ascontiguousarray(ref_xyb, F)
    dis_xyb = np.ascontiguousarray(dis_xyb, F)
    # img[0..2] = sigma11, sigma22, sigma12 (nppiMul); img[8], img[9] = ref, dis
    planes = [ref_xyb * ref_xyb, dis_xyb * dis_xyb, ref_xyb * dis_xyb, ref_xyb, dis_xyb]
    # first blur_pass_fused: (i0->i3, i1->i4, i2->i5, i8->i6, i9->i7): columns of the image
    pass1 = _blur_all(planes)
    # nppiTranspose x5 into imgt[0..4], second blur_pass_fused over the columns of the transposed images -> imgt[5..9]
    pass1_t = [np.ascontiguousarray(p.transpose(0, 2, 1)) for p in pass1]
    pass2 = _blur_all(pass1_t)
    # ref / dis transposed into imgt[0], imgt[1]; compute_error_maps(source=i0, distorted=i1, mu1=i8, mu2=i9, sigma11=i5, ...)
    src_t = np.ascontiguousarray(ref_xyb.transpose(0, 2, 1))
    dis_t = np.ascontiguousarray(dis_xyb.transpose(0, 2, 1))
    maps = error_maps(src_t, dis_t, pass2[3], pass2[4], pass2[0], pass2[1], pass2[2])
    # reduce(): nppiSum (Npp64f accumulators) of the map and of sqr_ip(sqr(map)) -- both squarings rounded to f32
    sums = np.zeros((6, 3), np.float64)
    for m in range(3):
        q = maps[m] * maps[m]
        q = q * q
        for c in range(3):
            sums[m, c] = float(np.sum(maps[m][c].astype(np.float64)))
            sums[3 + m, c] = float(np.sum(q[c].astype(np.float64)))
    return sums, {"pass1": pass1_t, "pass2": pass2, "maps": list(maps)}


def ssimulacra2_sums(ref_lin, dis_lin, cbrt="exact", capture=False):
    """record() (lib.rs:140-229): scale 0 from the inputs, scale s from downscale_by_2 of the LINEAR image of scale s-1.
    -> sums (6 scales, 6 kinds, 3 channels) [, list over scales of {xyb: (ref, dis), pass1, pass2, maps}]"""
    lin = [np.ascontiguousarray(ref_lin, F), np.ascontiguousarray(dis_lin, F)]
    sums = np.zeros((SCALES, 6, 3), np.float64)
    cap = []
    for scale in range(SCALES):
        if scale > 0:
            lin = [downscale_by_2(a) for a in lin]
        xyb = [linear_to_xyb(a, cbrt) for a in lin]
        sums[scale], inter = process_scale(xyb[0], xyb[1])
        if capture:
            inter["xyb"] = xyb
            cap.append(inter)
    return (sums, cap) if capture else sums


def weights():
    with open(os.path.join(_ROOT, "tests", "golden", "reference_tables.json")) as f:
        return np.array(json.load(f)["weights"], np.float64)


def score_from_sums(sums, width, height):
    """post_process_scores, lib.rs:586-622; sums (6, 6, 3) = scores[scale * 18 + kind * 3 + channel]"""
    W = weights()
    sc = np.array(sums, np.float64).reshape(-1).copy()
    w, h = width, height
    for scale in range(SCALES):
        opp = 1.0 / float(w * h)  # NppiRect::norm of the scale's size
        for c in range(3):
            o = 18 * scale + c
            ow = c * 36 + 6 * scale
            for k in range(3):
                sc[o + 3 * k] = abs(sc[o + 3 * k] * opp) * W[ow + k]
            for k in range(3, 6):
                sc[o + 3 * k] = math.sqrt(math.sqrt(sc[o + 3 * k] * opp)) * W[ow + k]
        w, h = (w + 1) // 2, (h + 1) // 2
    score = 0.0
    for v in sc:  # `self.scores.iter().sum()`: left to right
        score += float(v)
    score *= 0.9562382616834844
    # (6.2e-5 * s * s).mul_add(s, 2.32.mul_add(s, -0.0208 * s * s)) in f64: the two fused operations are evaluated with
    # Fractions and rounded once each
    from fractions import Fraction
    inner = float(Fraction(2.326765642916932) * Fraction(score) + Fraction(-0.020884521182843837 * score * score))
    score = float(Fraction(6.248496625763138e-5 * score * score) * Fraction(score) + Fraction(inner))
    if score > 0.0:
        score = float(Fraction(math.pow(score, 0.6276336467831387)) * Fraction(-10.0) + Fraction(100.0))
    else:
        score = 100.0
    return score


# ---------------------------------------------------------------------------------------------------------------------
# YUV 4:2:0 biplanar -> linear RGB, cuda-colorspace-kernel
# ---------------------------------------------------------------------------------------------------------------------
_PRIMARIES = {  # constants.rs:3-18: R, G, B, white (D65)
    0: ((0.640, 0.330), (0.300, 0.600), (0.150, 0.060), (0.3127, 0.3290)),  # BT709
    1: ((0.630, 0.340), (0.310, 0.595), (0.155, 0.070), (0.3127, 0.3290)),  # BT601_525
    2: ((0.640, 0.330), (0.290, 0.600), (0.150, 0.060), (0.3127, 0.3290)),  # BT601_625
}


def _xy_to_xyz(p):
    x, y = F(p[0]), F(p[1])
    return (x / y, F(1.0), ((F(1.0) - x) - y) / y)


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def kr_kb(matrix):
    """constants_from_primaries (lib.rs:203-218), float32 throughout (Rust const evaluation is IEEE)"""
    r, g, b, w = (_xy_to_xyz(p) for p in _PRIMARIES[matrix])
    x_rgb, y_rgb, z_rgb = (r[0], g[0], b[0]), (r[1], g[1], b[1]), (r[2], g[2], b[2])
    mul = F(1.0) / _dot(x_rgb, _cross(y_rgb, z_rgb))
    return _dot(w, _cross(g, b)) * mul, _dot(w, _cross(r, g)) * mul


def yuv_coefficients(matrix, bits):
    """MatrixCoefficients::coefficients::<Limited, Limited, N> (lib.rs:186-200): y, r, b, g1, g2"""
    kr, kb = kr_kb(matrix)
    one, two = F(1.0), F(2.0)
    luma_range = F((235 << (bits - 8)) - (16 << (bits - 8)))
    chroma_range = F((240 << (bits - 8)) - (16 << (bits - 8)))
    kg = (one - kr) - kb
    return (one / luma_range,
            ((two * (one - kr)) * one) / chroma_range,
            ((two * (one - kb)) * one) / chroma_range,
            ((((-two * (one - kb)) * kb) / kg) * one) / chroma_range,
            ((((-two * (one - kr)) * kr) / kg) * one) / chroma_range)


_BETA = F(0.018053968510807)
_ALPHA = F(F(1.0) + F(5.5) * _BETA)
_THRESHOLD = F(0.08124285829863521110029445797874)
_ALPHA_M1 = F(_ALPHA - F(1.0))
_INV_GAMMA = F(F(1.0) / F(0.45))


def bt709_eotf(v, mode="exact"):
    """BT709::eotf (lib.rs:221-236): v >= threshold ? powf_fast((v + (alpha - 1)) / alpha, 1 / 0.45) : v / 4.5"""
    v = np.asarray(v, F)
    base = (v + _ALPHA_M1) / _ALPHA
    safe = np.where(v >= _THRESHOLD, base, F(1.0))
    if mode == "exact":
        pw = np.power(safe.astype(np.float64), np.float64(_INV_GAMMA)).astype(F)
    elif mode == "fast_powf":
        pw = np.exp2((_INV_GAMMA * np.log2(safe).astype(F)).astype(F)).astype(F)
    else:
        raise ValueError(mode)
    return np.where(v >= _THRESHOLD, pw, v / F(4.5)).astype(F)


def yuv420_biplanar_to_linear(surface, pitch, coded_height, w, h, bits, matrix, eotf="exact"):
    """biplanaryuv420_to_linearrgb_generic (biplanar.rs:8-70) over the NVDEC surface contract (dec.rs:299-393): luma rows at
    `pitch` bytes, the CbCr plane at pitch * coded_height.  The kernel is launched over (w/2, h/2) quads
    (cuda-colorspace/src/kernel.rs:64-65): an odd last column / row is not written; it reads 0 here.  -> (3, h, w)"""
    tr = (lambda x: bt709_eotf(x, eotf)) if isinstance(eotf, str) else eotf
    dt = np.uint8 if bits == 8 else np.uint16
    surf = np.ascontiguousarray(surface, np.uint8)
    rows = surf.reshape(-1, pitch).view(dt)
    qw, qh = w // 2, h // 2
    Y = rows[:2 * qh, :2 * qw].astype(np.int64)
    UV = rows[coded_height:coded_height + qh, :2 * qw].astype(np.int64)
    neutral = 1 << (bits - 1)
    ymin = 16 << (bits - 8)
    ky, kr_, kb_, kg1, kg2 = yuv_coefficients(matrix, bits)
    cb = (UV[:, 0::2] - neutral).astype(F)
    cr = (UV[:, 1::2] - neutral).astype(F)
    r_ = kr_ * cr
    g_ = fma32(kg1, cb, kg2 * cr).reshape(cb.shape)
    b_ = kb_ * cb
    up = lambda a: np.repeat(np.repeat(a, 2, axis=0), 2, axis=1)
    luma = (np.maximum(Y, ymin) - ymin).astype(F) * ky
    out = np.zeros((3, h, w), F)
    for i, add in enumerate((r_, g_, b_)):
        out[i, :2 * qh, :2 * qw] = np.minimum(np.maximum(np.asarray(tr(luma + up(add)), F), F(0.0)), F(1.0))
    return out


def ssimulacra2_from_linear(ref_lin, dis_lin, cbrt="exact"):
    _, h, w = np.asarray(ref_lin).shape
    sums = ssimulacra2_sums(ref_lin, dis_lin, cbrt)
    return score_from_sums(sums, w, h), sums
