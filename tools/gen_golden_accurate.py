#!/usr/bin/env python3
"""Writes tests/golden/scores_accurate_frozen.json -- FROZEN: generated once from the most accurate evaluation of the reference's
expressions and never regenerated when the product's arithmetic changes (the regression goldens of
tools/gen_golden_scores.py follow the oracle, so they cannot bound drift from the upstream arithmetic).

Generator = the numpy twin (oracle/twin_numpy.py, written from the reference's files) with the two closed libdevice functions
replaced by correctly rounded ones: cbrtf -> float64 cbrt rounded once, __nv_fast_powf -> float64 pow of the reference's
f32-rounded base, rounded once.  Beside each accurate score the file records, for the same inputs,
  fast_powf_shape   the twin with the transfer function evaluated like libdevice's fast path, exp2f(y * log2f(x)) in f32
  build             the C oracle (= the HIP kernels, bit for bit) with the arithmetic of round `build_round`; `build_history`
                    keeps the figures of earlier versions (r02: transfer function fitted in v, f32; r03a: an f32 cubic on the reference's f32
                    base, <= 0.69 ulp; r03b: that base, binary64 cubic, correctly rounded, cube root <= 0.5003 ulp; r03: + the cube
                    root's sixth-order step, 11 of 25 M arguments not the nearest float)
  bound             max(2 x |build - accurate|, 1e-4): what tests/test_golden_accurate.py allows oracle and HIP path
so that the distance between this build and the reference's own arithmetic is a committed number, not prose.
`accurate` and `fast_powf_shape` are frozen; `--refresh-build` recomputes only the build's own columns (after a deliberate
change of the product's arithmetic, in the same commit as that change).
The inputs are regenerated from their seeds (turbo-metrics_amd/synth.py); only numbers are stored."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import twin_numpy as T  # noqa: E402
from tm_pkg import tm  # noqa: E402

CASES = [("nv12", 160, 96, 1, 0), ("nv12", 333, 203, 4, 1), ("nv12", 640, 360, 7, 0), ("p016", 320, 200, 2, 0), ("rgb8", 256, 192, 0, 0),
         ("nv12", 1920, 1080, 2, 0), ("rgb8", 1920, 1080, 0, 0)]
BUILD_ROUND = "r03"
FLOOR = 1e-4  # smallest per-case bound = north_star's tolerance
OUT = os.path.join(ROOT, "tests", "golden", "scores_accurate_frozen.json")


def srgb_lut():
    bits = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")))["srgb_lut_bits"]
    return np.array(bits, np.uint32).view(np.float32)


def twin_linear_pair(kind, w, h, n, matrix, eotf):
    if kind == "rgb8":  # srgb_to_linear_u8_lookup (cuda-colorspace-kernel/src/srgb.rs:51-66): the reference's own table
        lut = srgb_lut()
        r8, d8 = tm.synth.rgb8_pair(w, h)
        return (np.ascontiguousarray(lut[r8].transpose(2, 0, 1)), np.ascontiguousarray(lut[d8].transpose(2, 0, 1)))
    gen = tm.synth.nv12_pair if kind == "nv12" else tm.synth.p016_pair
    (rs, rp, rch), (ds, dp, dch) = gen(w, h, n)
    bits = 8 if kind == "nv12" else 16
    return (T.yuv420_biplanar_to_linear(rs, rp, rch, w, h, bits, matrix, eotf=eotf),
            T.yuv420_biplanar_to_linear(ds, dp, dch, w, h, bits, matrix, eotf=eotf))


def oracle_score(kind, w, h, n, matrix):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden_scores as G
    lr, ld = G.linear_pair(kind, w, h, n, matrix)
    return O.ssimulacra2_from_linear(lr, ld)[0]


def compute_case(kind, w, h, n, matrix, with_fast=True):
    out = {"kind": kind, "width": w, "height": h, "pair": n, "matrix": matrix}
    lr, ld = twin_linear_pair(kind, w, h, n, matrix, "exact")
    out["accurate"] = T.ssimulacra2_from_linear(lr, ld, cbrt="exact")[0]
    if with_fast and kind != "rgb8":
        lr, ld = twin_linear_pair(kind, w, h, n, matrix, "fast_powf")
        out["fast_powf_shape"] = T.ssimulacra2_from_linear(lr, ld, cbrt="exact")[0]
    return out


def bound_of(dev):
    return max(2.0 * abs(dev), FLOOR)


def refresh_build():
    doc = json.load(open(OUT))
    for c in doc["cases"]:
        hist = c.setdefault("build_history", {})
        if "oracle_at_freeze" in c:  # round-2 layout
            hist["r02"] = c.pop("oracle_at_freeze") - c["accurate"]
            c.pop("oracle_minus_accurate", None)
        elif c.get("build_round") not in (None, BUILD_ROUND) or "--keep-as" in sys.argv:
            hist[sys.argv[sys.argv.index("--keep-as") + 1] if "--keep-as" in sys.argv else c["build_round"]] = c["build_minus_accurate"]
        c["build"] = oracle_score(c["kind"], c["width"], c["height"], c["pair"], c["matrix"])
        c["build_round"] = BUILD_ROUND
        c["build_minus_accurate"] = c["build"] - c["accurate"]
        c["bound"] = bound_of(c["build_minus_accurate"])
        print(c, flush=True)
    doc.pop("band", None)
    doc["floor"] = FLOOR
    json.dump(doc, open(OUT, "w"), indent=1)


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if "--refresh-build" in sys.argv:
        return refresh_build()
    if os.path.exists(OUT) and "--force" not in sys.argv:
        raise SystemExit(f"{OUT} exists and is FROZEN; pass --force only to add cases, never because the product's arithmetic changed")
    cases = []
    for kind, w, h, n, matrix in CASES:
        c = compute_case(kind, w, h, n, matrix)
        c["build"] = oracle_score(kind, w, h, n, matrix)
        c["build_round"] = BUILD_ROUND
        c["build_minus_accurate"] = c["build"] - c["accurate"]
        c["bound"] = bound_of(c["build_minus_accurate"])
        if "fast_powf_shape" in c:
            c["fast_powf_shape_minus_accurate"] = c["fast_powf_shape"] - c["accurate"]
        print(c, flush=True)
        cases.append(c)
    json.dump({"generator": "tools/gen_golden_accurate.py", "frozen": True, "floor": FLOOR, "cases": cases}, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
