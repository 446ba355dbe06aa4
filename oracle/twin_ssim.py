"""Second, independent statement of SSIM and MS-SSIM -- TEST INFRASTRUCTURE ONLY (never imported by the product).

Written from the two papers, in float64 numpy/scipy, without reference to oracle/tm_ssim.c (the C statement the HIP
kernels are held to). tests/test_ssim_twin.py holds the two against each other, so that the build-defined A5 metrics
(NPP's nppiSSIM / nppiWMSSSIM are closed source, reference call sites
crates/turbo-metrics/src/lib.rs:308-339 and crates/cudarse/cudarse-npp/src/image/ist.rs:106-179) have two witnesses.

  [W04] Z. Wang, A. Bovik, H. Sheikh, E. Simoncelli, "Image quality assessment: from error visibility to structural
        similarity", IEEE TIP 13(4), 2004.  Eq. (13) with the 11x11 circular-symmetric Gaussian window, sigma 1.5,
        normalised to unit sum (section III-C); K1 = 0.01, K2 = 0.03, L = 255; mean over all window positions that lie
        inside the image (eq. 17).
  [W03] Z. Wang, E. Simoncelli, A. Bovik, "Multi-scale structural similarity for image quality assessment", Asilomar
        2003.  Eq. (7): l_M^aM * prod_j c_j^bj s_j^gj with bj = gj = aj = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333),
        M = 5; scale j+1 = low-pass filter + downsample by 2 of scale j.  As in the authors' released implementation
        the low-pass filter is the 2x2 mean, contrast*structure is one term cs (C3 = C2/2), and each scale contributes
        the MEAN of its map (cs on scales 1-4, l*cs on scale 5).

Input here = the pair the reference hands to NPP: linear RGB in [0,1] quantised to u8 with round-half-even
(cuda-colorspace-kernel/src/sample_conv.rs:6-35), per channel, then the mean of the three channels.

Choices the papers leave open (DESIGN.md section 4 lists them with the choice made):
  * odd width/height at a decimation step: `odd="drop"` keeps floor(n/2) complete 2x2 blocks, `odd="clamp"` keeps
    ceil(n/2) with the last block's missing sample replicated (symmetric boundary of the authors' imfilter call);
  * variances as E[x^2] - mu^2 over the same window (eq. 14-16 with the window weights; the unbiased N-1 variant of
    the text's unweighted formula does not apply to a weighted window).
"""
import numpy as np
from scipy.signal import correlate2d

K1, K2, L = 0.01, 0.03, 255.0
C1, C2 = (K1 * L) ** 2, (K2 * L) ** 2
EXPONENTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
WIN, SIGMA = 11, 1.5


def window():
    r = np.arange(WIN, dtype=np.float64) - (WIN - 1) / 2
    g = np.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2.0 * SIGMA * SIGMA))
    return g / g.sum()


def quantize(lin):
    """linear [0,1] float -> the u8 sample NPP sees, as float64"""
    v = np.rint(np.asarray(lin, np.float32) * np.float32(255.0))  # np.rint: half to even, like float2uint_rn
    return np.clip(v, 0.0, 255.0).astype(np.float64)


def _local(x, w):
    return correlate2d(x, w, mode="valid")


def maps(x, y):
    """(ssim map, cs map) over the valid window positions of one channel; x, y float64 (h, w), h and w >= 11"""
    w = window()
    mx, my = _local(x, w), _local(y, w)
    sxx = _local(x * x, w) - mx * mx
    syy = _local(y * y, w) - my * my
    sxy = _local(x * y, w) - mx * my
    cs = (2.0 * sxy + C2) / (sxx + syy + C2)
    lum = (2.0 * mx * my + C1) / (mx * mx + my * my + C1)
    return lum * cs, cs


def halve(x, odd):
    h, w = x.shape
    if odd == "clamp":
        if h & 1:
            x = np.concatenate([x, x[-1:]], axis=0)
        if w & 1:
            x = np.concatenate([x, x[:, -1:]], axis=1)
    else:
        x = x[: h & ~1, : w & ~1]
    return 0.25 * (x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2])


def scale_means(ref_lin, dis_lin, n_scales=5, odd="drop"):
    """(3, n_scales, 2) float64: mean ssim and mean cs of every channel at every scale; inputs (3, h, w) linear RGB.
    Scales that no longer hold one window are NaN."""
    out = np.full((3, n_scales, 2), np.nan)
    for c in range(3):
        x, y = quantize(ref_lin[c]), quantize(dis_lin[c])
        for s in range(n_scales):
            if min(x.shape) < WIN:
                break
            sm, cm = maps(x, y)
            out[c, s] = sm.mean(), cm.mean()
            x, y = halve(x, odd), halve(y, odd)
    return out


def ssim(ref_lin, dis_lin):
    return float(scale_means(ref_lin, dis_lin, 1)[:, 0, 0].mean())


def msssim(ref_lin, dis_lin, odd="drop"):
    m = scale_means(ref_lin, dis_lin, 5, odd)
    e = np.asarray(EXPONENTS)
    per_channel = np.prod(m[:, :4, 1] ** e[:4], axis=1) * m[:, 4, 0] ** e[4]
    return float(per_channel.mean())
