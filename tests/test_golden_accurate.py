"""FROZEN goldens (tests/golden/scores_accurate_frozen.json, tools/gen_golden_accurate.py): SSIMULACRA2 scores of the seeded
synthetic pairs from the most accurate evaluation of the reference's expressions (numpy twin, correctly rounded cbrt and pow).
They are never regenerated when the product's arithmetic changes; oracle and HIP path must stay within BAND of them.

What the band means: the reference calls closed libdevice code (__nv_fast_powf ~ exp2f(y log2f x), __nv_cbrtf 1 ulp); the file
also holds the score of a fast_powf-SHAPED evaluation of the same inputs, which lands 3e-4 ... 1.1e-2 away from `accurate` --
the same order as this build's own distance (2e-5 ... 2.2e-2).  North_star's 1e-4 is held between HIP and oracle (bit-identical
planes); against upstream's bits the honest statement is this band (the reference's own GPU-vs-CPU check allows 0.25)."""
import json
import os
import sys

import pytest

from tm_pkg import tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_accurate as GA  # noqa: E402

DOC = json.load(open(os.path.join(ROOT, "tests", "golden", "scores_accurate_frozen.json")))
CASES, BAND = DOC["cases"], DOC["band"]
ID = lambda c: f'{c["kind"]}_{c["width"]}x{c["height"]}'  # noqa: E731


def test_file_is_frozen_and_the_committed_deviation_figures_hold():
    assert DOC["frozen"] is True and BAND == 5e-2 and len(CASES) == 7
    for c in CASES:
        assert abs(c["oracle_minus_accurate"]) <= BAND
        if "fast_powf_shape" in c:  # what libdevice's fast path alone would move the score by: same order as this build's distance
            assert 1e-4 < abs(c["fast_powf_shape_minus_accurate"]) <= BAND
    assert max(abs(c["oracle_minus_accurate"]) for c in CASES) > 1e-2  # ... which is why 1e-4 against upstream is not claimed


@pytest.mark.parametrize("case", [c for c in CASES if c["width"] <= 333], ids=ID)
def test_twin_reproduces_the_frozen_scores(case):
    got = GA.compute_case(case["kind"], case["width"], case["height"], case["pair"], case["matrix"], with_fast=case["width"] <= 160)
    assert abs(got["accurate"] - case["accurate"]) <= 1e-9
    if "fast_powf_shape" in got:
        assert abs(got["fast_powf_shape"] - case["fast_powf_shape"]) <= 1e-9


@pytest.mark.parametrize("case", [c for c in CASES if c["width"] * c["height"] <= 640 * 360], ids=ID)
def test_oracle_stays_within_the_band_of_the_frozen_scores(case):
    got = GA.oracle_score(case["kind"], case["width"], case["height"], case["pair"], case["matrix"])
    assert abs(got - case["accurate"]) <= BAND


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=ID)
def test_hip_stays_within_the_band_of_the_frozen_scores(case):
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
    assert abs(s.ssimulacra2 - case["accurate"]) <= BAND
