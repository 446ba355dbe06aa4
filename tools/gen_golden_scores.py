#!/usr/bin/env python3
"""Writes tests/golden/scores_regression.json: scores of seeded synthetic frame pairs computed by the CPU oracle
(oracle/tm_oracle.c, oracle/tm_ssim.c, oracle/tm_cpu_path.c).  The inputs are regenerated from their seeds by the tests
(turbo-metrics_amd/synth.py); only the expected numbers are stored.  These are regression fixtures of THIS build's
oracle -- the reference ships no golden vector for this path -- checked at the north-star tolerance (1e-4),
so that a change of the arithmetic on both sides of the parity tests at once cannot go unnoticed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tm_pkg import tm  # noqa: E402

CASES = [("nv12", 160, 96, 1, 0), ("nv12", 333, 203, 4, 1), ("nv12", 640, 360, 7, 0), ("p016", 320, 200, 2, 0), ("rgb8", 256, 192, 0, 0),
         ("nv12", 1920, 1080, 2, 0),
         ("rgb8", 1920, 1080, 0, 0)]  # BASELINE config 1: single 1080p RGB8 (PNG) pair; `cpu_path_ssimulacra2` = the reference's CPU path restated


def linear_pair(kind, w, h, n, matrix):
    if kind == "rgb8":
        r8, d8 = tm.synth.rgb8_pair(w, h)
        return O.rgb8_to_linear(r8), O.rgb8_to_linear(d8)
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    (rs, rp, rch), (ds, dp, dch) = gen(w, h, n)
    bits = 8 if kind == "nv12" else 16
    return O.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, matrix), O.yuv420_biplanar_to_linear(ds, dp, dch, w, h, bits, matrix)


def main():
    out = []
    for kind, w, h, n, matrix in CASES:
        lr, ld = linear_pair(kind, w, h, n, matrix)
        score, sums = O.ssimulacra2_from_linear(lr, ld)
        sse, psnr = O.psnr(lr, ld)
        ssim, msssim, _ = O.ssim_msssim(lr, ld)
        out.append({"kind": kind, "width": w, "height": h, "pair": n, "matrix": matrix, "ssimulacra2": score, "sse": sse, "psnr": psnr,
                    "ssim": ssim, "msssim": None if np.isnan(msssim) else msssim, "cpu_path_ssimulacra2": O.cpu_path_score_linear(lr, ld),
                    "sum_of_raw_sums": float(np.sum(sums))})
        print(out[-1])
    with open(os.path.join(ROOT, "tests", "golden", "scores_regression.json"), "w") as f:
        json.dump({"generator": "tools/gen_golden_scores.py", "arithmetic": "r03 (BT.709 transfer function on the reference's f32 base)", "cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
