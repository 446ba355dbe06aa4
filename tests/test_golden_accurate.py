"""FROZEN goldens (tests/golden/scores_accurate_frozen.json, tools/gen_golden_accurate.py): SSIMULACRA2 scores of the seeded
synthetic pairs from the most accurate evaluation of the reference's expressions (numpy twin, correctly rounded cbrt and pow).
`accurate` is never regenerated when the product's arithmetic changes; oracle and HIP path must stay within each case's own
`bound` = max(2 x the committed |build - accurate|, 1e-4) of it (round 2: one flat band of 5e-2).

What the figures mean: the reference calls closed libdevice code (__nv_fast_powf ~ exp2f(y log2f x), __nv_cbrtf 1 ulp); the file
also holds the score of a fast_powf-SHAPED evaluation of the same inputs, which lands 3e-4 ... 1.1e-2 away from `accurate`.
Since round 3 both stand-ins are the reference's expressions CORRECTLY ROUNDED but for a handful of arguments (transfer function:
the reference's f32 base, a binary64 cubic, one rounding, 117 of 15.4 M arguments not the nearest float; cube root: 11 of 25 M),
so this build reproduces `accurate` itself: four of the seven cases to the last bit of the f64 score, the others within 4.3e-5 --
inside north_star's 1e-4.  `build_history` keeps the road there: r02 2e-5 ... 2.1e-2 (transfer function fitted in v), r03a
5e-4 ... 1.8e-2 (an f32 cubic within 0.69 ulp everywhere -- the 1080p NV12 case moves by 7e-3 ... 4e-2 when a random 0.8 % of its
linear samples move by ONE ulp, tools/score_conditioning.py), r03b <= 2.4e-3 (transfer function correctly rounded, cube root
0.5003 ulp).  What stays out of reach is libdevice's own deviation from these expressions (second column of DESIGN.md section 4)."""
import json
import os
import sys

import pytest

from tm_pkg import tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_accurate as GA  # noqa: E402

DOC = json.load(open(os.path.join(ROOT, "tests", "golden", "scores_accurate_frozen.json")))
CASES = DOC["cases"]
ID = lambda c: f'{c["kind"]}_{c["width"]}x{c["height"]}'  # noqa: E731


def test_file_is_frozen_and_the_committed_deviation_figures_hold():
    assert DOC["frozen"] is True and DOC["floor"] == 1e-4 and len(CASES) == 7
    for c in CASES:
        assert c["build_round"] == GA.BUILD_ROUND
        assert c["bound"] == max(2.0 * abs(c["build_minus_accurate"]), 1e-4)
        if "fast_powf_shape" in c:  # what libdevice's fast path alone would move the score by
            assert 1e-4 < abs(c["fast_powf_shape_minus_accurate"]) <= 5e-2
    assert max(abs(c["build_minus_accurate"]) for c in CASES) <= 1e-4  # all seven inside north_star's tolerance of the accurate evaluation
    assert sum(c["build_minus_accurate"] == 0.0 for c in CASES) >= 4  # ... four of them to the last bit
    # round 3 moved every YUV case towards `accurate` (the RGB8 cases do not use the transfer function)
    for c in CASES:
        assert abs(c["build_minus_accurate"]) <= abs(c["build_history"]["r02"])


@pytest.mark.parametrize("case", [c for c in CASES if c["width"] <= 333], ids=ID)
def test_twin_reproduces_the_frozen_scores(case):
    got = GA.compute_case(case["kind"], case["width"], case["height"], case["pair"], case["matrix"], with_fast=case["width"] <= 160)
    assert abs(got["accurate"] - case["accurate"]) <= 1e-9
    if "fast_powf_shape" in got:
        assert abs(got["fast_powf_shape"] - case["fast_powf_shape"]) <= 1e-9


@pytest.mark.parametrize("case", [c for c in CASES if c["width"] * c["height"] <= 640 * 360], ids=ID)
def test_oracle_stays_within_its_bound_of_the_frozen_scores(case):
    got = GA.oracle_score(case["kind"], case["width"], case["height"], case["pair"], case["matrix"])
    assert abs(got - case["accurate"]) <= case["bound"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ID)
def test_hip_stays_within_its_bound_of_the_frozen_scores(case):
    tm.init_hip(0)
    w, h, kind = case["width"], case["height"], case["kind"]
    if kind == "rgb8":
        r8, d8 = tm.synth.rgb8_pair(w, h)
        fr, fd = tm.HwFrame.rgb(r8), tm.HwFrame.rgb(d8)
    else:
        gen, mk = (tm.synth.nv12_pair, tm.HwFrame.nv12) if kind == "nv12" else (tm.synth.p016_pair, tm.HwFrame.p016)
        (rs, rp, rch), (ds, dp, dch) = gen(w, h, case["pair"])
        fr, fd = mk(rs, rp, rch, tm.ColorMatrix(case["matrix"])), mk(ds, dp, dch, tm.ColorMatrix(case["matrix"]))
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=1)
    s = eng.compute_one(fr, fd)
    eng.close()
    assert abs(s.ssimulacra2 - case["accurate"]) <= case["bound"]
