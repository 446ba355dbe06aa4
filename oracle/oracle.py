"""ctypes front end of the CPU parity oracle (oracle/tm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (turbo-metrics_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtm_oracle.so")
SCALES = 6


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("tm_oracle.c", "tm_cpu_path.c", "tm_ssim.c", "tm_math.h", "tm_oracle_tables.inc", "tm_math_tables.inc", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        L = _lib
        fp, dp, u8p, vp = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_void_p
        L.tmo_math_cbrtf.restype = C.c_float; L.tmo_math_cbrtf.argtypes = [C.c_float]
        L.tmo_math_powf.restype = C.c_float; L.tmo_math_powf.argtypes = [C.c_float, C.c_float]
        L.tmo_srgb_inverse_oetf.restype = C.c_float; L.tmo_srgb_inverse_oetf.argtypes = [C.c_float]
        L.tmo_bt709_eotf.restype = C.c_float; L.tmo_bt709_eotf.argtypes = [C.c_float]
        L.tmo_bt709_eotf_max_ulp2.restype = C.c_double; L.tmo_bt709_eotf_max_ulp2.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_long)]
        L.tmo_psnr_from_sse.restype = C.c_double; L.tmo_psnr_from_sse.argtypes = [C.c_uint64, C.c_size_t]
        L.tmo_sse_u8.restype = C.c_uint64; L.tmo_sse_u8.argtypes = [vp, vp, C.c_size_t]
        L.tmo_score_from_sums.restype = C.c_double; L.tmo_score_from_sums.argtypes = [dp, C.c_int, C.c_int]
        L.tmo_ssimulacra2_from_linear.restype = C.c_double
        L.tmo_ssimulacra2_from_linear.argtypes = [fp, fp, C.c_int, C.c_int, dp]
        L.tmo_cpu_path_score_linear.restype = C.c_double
        L.tmo_cpu_path_score_linear.argtypes = [fp, fp, C.c_int, C.c_int]
        L.tmo_cpu_path_run.restype = C.c_double
        L.tmo_cpu_path_run.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.tmo_cpu_path_score_srgb8.restype = C.c_double
        L.tmo_cpu_path_score_srgb8.argtypes = [vp, vp, C.c_int, C.c_int]
        L.tmo_ssim_from_sums.restype = C.c_double; L.tmo_ssim_from_sums.argtypes = [dp, C.c_int, C.c_int]
        L.tmo_msssim_from_sums.restype = C.c_double; L.tmo_msssim_from_sums.argtypes = [dp, C.c_int, C.c_int]
        L.tmo_yuv420_biplanar_to_linear.restype = C.c_int
        L.tmo_yuv420_biplanar_to_linear.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, fp]
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def scale_sizes(w, h):
    out = [(w, h)]
    for _ in range(1, SCALES):
        w, h = (w + 1) // 2, (h + 1) // 2
        out.append((w, h))
    return out


def srgb8_lut():
    out = np.zeros(256, np.float32)
    lib().tmo_srgb8_lut(_fp(out))
    return out


def weights():
    out = np.zeros(108, np.float64)
    lib().tmo_weights(_dp(out))
    return out


def gaussian_constants():
    out = np.zeros(7, np.float32)
    lib().tmo_gaussian_constants(_fp(out))
    return out


def cbrtf(a):
    L = lib()
    a = np.asarray(a, np.float32)
    return np.array([L.tmo_math_cbrtf(float(v)) for v in a.ravel()], np.float32).reshape(a.shape)


def powf(x, y):
    L = lib()
    x = np.asarray(x, np.float32)
    return np.array([L.tmo_math_powf(float(v), float(y)) for v in x.ravel()], np.float32).reshape(x.shape)


def bt709_eotf(v):
    L = lib()
    v = np.asarray(v, np.float32)
    return np.array([L.tmo_bt709_eotf(float(x)) for x in v.ravel()], np.float32).reshape(v.shape)


def cbrtf_scan(lo, hi):
    """(largest error in ulps, number of not correctly rounded results) over every float of [lo, hi)"""
    L = lib()
    L.tmo_cbrtf_scan.restype = C.c_double; L.tmo_cbrtf_scan.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_longlong)]
    n = C.c_longlong()
    return float(L.tmo_cbrtf_scan(float(lo), float(hi), C.byref(n))), int(n.value)


def bt709_eotf_max_ulp():
    """(largest error in ulps of the reference's expression -- its f32 base, exact pow -- over every float of the power branch,
    the argument where it occurs, the number of arguments whose result is not the correctly rounded value)"""
    w, n = C.c_float(), C.c_long()
    return float(lib().tmo_bt709_eotf_max_ulp2(C.byref(w), C.byref(n))), float(w.value), int(n.value)


def kr_kb(matrix):
    kr, kb = C.c_float(), C.c_float()
    lib().tmo_kr_kb(int(matrix), C.byref(kr), C.byref(kb))
    return np.float32(kr.value), np.float32(kb.value)


def yuv_coefficients(matrix, bits):
    out = np.zeros(5, np.float32)
    lib().tmo_yuv_coefficients(int(matrix), int(bits), _fp(out))
    return out


def rgb8_to_linear(rgb):
    """rgb: (h, w, 3) uint8 -> (3, h, w) float32 linear."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros((3, h, w), np.float32)
    lib().tmo_rgb8_to_linear(rgb.ctypes.data_as(C.c_void_p), C.c_size_t(w * 3), w, h, _fp(out))
    return out


def rgb16_to_linear(rgb):
    rgb = np.ascontiguousarray(rgb, np.uint16)
    h, w, _ = rgb.shape
    out = np.zeros((3, h, w), np.float32)
    lib().tmo_rgb16_to_linear(rgb.ctypes.data_as(C.c_void_p), C.c_size_t(w * 6), w, h, _fp(out))
    return out


def rgbf32_to_linear(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    out = np.zeros((3, h, w), np.float32)
    lib().tmo_rgbf32_to_linear(rgb.ctypes.data_as(C.c_void_p), C.c_size_t(w * 12), w, h, _fp(out))
    return out


def linear_packed_to_planar(rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    out = np.zeros((3, h, w), np.float32)
    lib().tmo_linear_packed_to_planar(rgb.ctypes.data_as(C.c_void_p), C.c_size_t(w * 12), w, h, _fp(out))
    return out


def yuv420_biplanar_to_linear(surface, pitch, coded_height, w, h, bits, matrix):
    """surface: 1-D uint8 buffer laid out like an NVDEC mapping (reference
    cudarse-video/src/dec.rs:299-393): luma rows at pitch, then the interleaved CbCr plane at
    pitch*coded_height.  Returns (3, h, w) float32 linear RGB."""
    surface = np.ascontiguousarray(surface, np.uint8)
    out = np.zeros((3, h, w), np.float32)
    base = surface.ctypes.data
    rc = lib().tmo_yuv420_biplanar_to_linear(C.c_void_p(base), C.c_void_p(base + pitch * coded_height),
                                             C.c_size_t(pitch), w, h, bits, matrix, _fp(out))
    if rc:
        raise ValueError("unsupported yuv configuration")
    return out


def quantize_u8(lin):
    lin = np.ascontiguousarray(lin, np.float32)
    out = np.zeros(lin.shape, np.uint8)
    lib().tmo_quantize_u8(_fp(lin), C.c_size_t(lin.size), out.ctypes.data_as(C.c_void_p))
    return out


def psnr(ref_lin, dis_lin):
    a, b = quantize_u8(ref_lin), quantize_u8(dis_lin)
    sse = lib().tmo_sse_u8(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_size_t(a.size))
    return int(sse), float(lib().tmo_psnr_from_sse(C.c_uint64(sse), C.c_size_t(a.size)))


def downscale_by_2(plane):
    plane = np.ascontiguousarray(plane, np.float32)
    h, w = plane.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().tmo_downscale_by_2(_fp(plane), w, h, _fp(out))
    return out


def linear_to_xyb(lin):
    lin = np.ascontiguousarray(lin, np.float32)
    out = np.zeros_like(lin)
    lib().tmo_linear_to_xyb(_fp(lin), C.c_size_t(lin[0].size), _fp(out))
    return out


def blur_columns(plane):
    plane = np.ascontiguousarray(plane, np.float32)
    h, w = plane.shape
    out = np.zeros_like(plane)
    lib().tmo_blur_columns(_fp(plane), w, h, _fp(out))
    return out


def error_maps(src, dis, mu1, mu2, s11, s22, s12):
    arrs = [np.ascontiguousarray(a, np.float32) for a in (src, dis, mu1, mu2, s11, s22, s12)]
    outs = [np.zeros_like(arrs[0]) for _ in range(3)]
    lib().tmo_error_maps(*[_fp(a) for a in arrs], C.c_size_t(arrs[0].size), *[_fp(o) for o in outs])
    return outs


def process_scale(ref_xyb, dis_xyb, capture=False):
    """ref_xyb, dis_xyb: (3, h, w).  Returns (sums18 [kind][channel], cap or None) where cap is a
    dict of TRANSPOSED-orientation planes (w rows, h columns): pass1[p][c], pass2[p][c], maps[m][c]."""
    ref_xyb = np.ascontiguousarray(ref_xyb, np.float32)
    dis_xyb = np.ascontiguousarray(dis_xyb, np.float32)
    _, h, w = ref_xyb.shape
    sums = np.zeros(18, np.float64)
    cap = np.zeros((13, 3, w, h), np.float32) if capture else None
    lib().tmo_process_scale(_fp(ref_xyb), _fp(dis_xyb), w, h, _dp(sums), _fp(cap) if capture else None)
    if capture:
        return sums.reshape(6, 3), {"pass1": cap[0:5], "pass2": cap[5:10], "maps": cap[10:13]}
    return sums.reshape(6, 3), None


def ssimulacra2_sums(ref_lin, dis_lin, want_xyb=False):
    """ref_lin, dis_lin: (3, h, w) linear RGB.  Returns sums (6 scales, 6 kinds, 3 channels) and,
    optionally, the XYB pyramid as a list over scales of (ref_xyb, dis_xyb) arrays (3, hs, ws)."""
    ref_lin = np.ascontiguousarray(ref_lin, np.float32)
    dis_lin = np.ascontiguousarray(dis_lin, np.float32)
    _, h, w = ref_lin.shape
    sums = np.zeros(108, np.float64)
    sizes = scale_sizes(w, h)
    total = sum(6 * ws * hs for ws, hs in sizes)
    xyb = np.zeros(total, np.float32) if want_xyb else None
    lib().tmo_ssimulacra2_sums(_fp(ref_lin), _fp(dis_lin), w, h, _dp(sums), _fp(xyb) if want_xyb else None)
    if not want_xyb:
        return sums.reshape(6, 6, 3)
    pyr, o = [], 0
    for ws, hs in sizes:
        n = 3 * ws * hs
        pyr.append((xyb[o:o + n].reshape(3, hs, ws), xyb[o + n:o + 2 * n].reshape(3, hs, ws)))
        o += 2 * n
    return sums.reshape(6, 6, 3), pyr


def score_from_sums(sums, w, h):
    sums = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    assert sums.size == 108
    return float(lib().tmo_score_from_sums(_dp(sums), w, h))


def ssimulacra2_from_linear(ref_lin, dis_lin):
    ref_lin = np.ascontiguousarray(ref_lin, np.float32)
    dis_lin = np.ascontiguousarray(dis_lin, np.float32)
    _, h, w = ref_lin.shape
    sums = np.zeros(108, np.float64)
    s = lib().tmo_ssimulacra2_from_linear(_fp(ref_lin), _fp(dis_lin), w, h, _dp(sums))
    return float(s), sums.reshape(6, 6, 3)


def cpu_path_score_linear(ref_lin, dis_lin):
    """Restated reference CPU path (examples/cpu.rs) on planar linear RGB (3, h, w)."""
    ref_lin = np.ascontiguousarray(ref_lin, np.float32)
    dis_lin = np.ascontiguousarray(dis_lin, np.float32)
    _, h, w = ref_lin.shape
    return float(lib().tmo_cpu_path_score_linear(_fp(ref_lin), _fp(dis_lin), w, h))


def cpu_path_run(surfaces, w, h, bits, n_pairs, n_threads):
    """Frame-level parallel run of the restated reference CPU path (tmo_cpu_path_run): `surfaces` = [((ref bytes, pitch, coded
    height), (dis bytes, pitch, coded height)), ...] decoded 4:2:0 biplanar surfaces (pair i of the run takes entry i % len);
    every worker thread converts its pair to linear RGB and scores it, buffers allocated once per worker.  Returns (seconds, scores)."""
    refs = [np.ascontiguousarray(np.asarray(r[0]).view(np.uint8)) for r, _ in surfaces]
    diss = [np.ascontiguousarray(np.asarray(d[0]).view(np.uint8)) for _, d in surfaces]
    pitch, coded_h = int(surfaces[0][0][1]), int(surfaces[0][0][2])
    assert all(int(r[1]) == pitch and int(r[2]) == coded_h and int(d[1]) == pitch and int(d[2]) == coded_h for r, d in surfaces)
    pa = (C.c_void_p * len(refs))(*[a.ctypes.data for a in refs])
    pb = (C.c_void_p * len(diss))(*[a.ctypes.data for a in diss])
    scores = np.zeros(n_pairs, np.float64)
    secs = float(lib().tmo_cpu_path_run(pa, pb, len(refs), pitch, coded_h, int(w), int(h), int(bits), int(n_pairs), int(n_threads), _dp(scores)))
    if secs < 0:
        raise RuntimeError("tmo_cpu_path_run failed (out of memory or threads)")
    return secs, scores


def cpu_path_score_srgb8(ref_rgb, dis_rgb):
    """Restated reference CPU path on packed sRGB u8 images (h, w, 3): CpuImg::from_srgb + compute_frame_ssimulacra2."""
    ref_rgb = np.ascontiguousarray(ref_rgb, np.uint8)
    dis_rgb = np.ascontiguousarray(dis_rgb, np.uint8)
    h, w, _ = ref_rgb.shape
    return float(lib().tmo_cpu_path_score_srgb8(ref_rgb.ctypes.data_as(C.c_void_p), dis_rgb.ctypes.data_as(C.c_void_p), w, h))


# ---- SSIM / MS-SSIM of the u8-quantised linear RGB pair (oracle/tm_ssim.c: BUILD-DEFINED, parity unpinned) ----------
def ssim_window():
    out = np.zeros(11, np.float32)
    lib().tmo_ssim_window(_fp(out))
    return out


def msssim_sums(ref_lin, dis_lin):
    """(3 channels, 5 scales, [sum ssim, sum cs]) of the quantised pair; ref_lin/dis_lin: (3, h, w) linear RGB"""
    a, b = quantize_u8(ref_lin), quantize_u8(dis_lin)
    _, h, w = a.shape
    sums = np.zeros(30, np.float64)
    lib().tmo_msssim_sums(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), w, h, _dp(sums))
    return sums.reshape(3, 5, 2)


def ssim_from_sums(sums, w, h):
    s = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    return float(lib().tmo_ssim_from_sums(_dp(s), w, h))


def msssim_from_sums(sums, w, h):
    s = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    return float(lib().tmo_msssim_from_sums(_dp(s), w, h))


def ssim_msssim(ref_lin, dis_lin):
    _, h, w = np.asarray(ref_lin).shape
    s = msssim_sums(ref_lin, dis_lin)
    return ssim_from_sums(s, w, h), msssim_from_sums(s, w, h), s
