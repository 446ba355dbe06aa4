#!/usr/bin/env python3
"""How much does the SSIMULACRA2 score move when the transfer function's last bits change?  (CPU, oracle only.)
Perturbs the linear RGB of a synthetic 640x360 pair by a few ulps, (a) independently per sample, (b) as a function of the value
(equal inputs stay equal -- what a different but consistent powf / cbrtf rounding does), and prints the score differences.
Result on this build: 1e-3 ... 2e-2 either way -- the clamp max(ssim, 0) rectifies rounding noise wherever ref and dis are (nearly)
equal -- so no restatement of the reference's closed-source fast_powf / cbrtf can be expected to reproduce its scores to 1e-4;
the reference's own GPU-vs-CPU check uses +-0.25 (ssimulacra2-cuda/examples/compare.rs:70-90).  DESIGN.md section 4."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from tm_pkg import tm

w, h = 640, 360
(rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, 7)
a = O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, 8, 0)
b = O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, 8, 0)


def score(x, y):
    sums, _ = O.ssimulacra2_sums(x, y, want_xyb=True)
    return float(O.score_from_sums(sums, w, h))


rng = np.random.default_rng(0)


def independent(x, ulps):
    x = np.array(x, np.float32, copy=True)
    y = (x.view(np.int32) + rng.integers(-ulps, ulps + 1, x.shape).astype(np.int32)).view(np.float32)
    return np.where((x > 0) & (x < 1), y, x).astype(np.float32)


def by_value(x, ulps, seed):
    x = np.array(x, np.float32, copy=True)
    xi = x.view(np.int32)
    k = ((xi.astype(np.int64) * 2654435761 + seed * 97) >> 7) % (2 * ulps + 1) - ulps
    return np.where((x > 0) & (x < 1), (xi + k.astype(np.int32)).view(np.float32), x).astype(np.float32)


s0 = score(a, b)
print("score", s0, "| bit-identical ref/dis samples:", float((np.asarray(a) == np.asarray(b)).mean()))
for u in (1, 2, 8):
    print(u, "ulp, independent :", [round(score(independent(a, u), independent(b, u)) - s0, 6) for _ in range(3)])
    print(u, "ulp, by value    :", [round(score(by_value(a, u, s), by_value(b, u, s)) - s0, 6) for s in range(3)])
