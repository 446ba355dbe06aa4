#!/usr/bin/env python3
"""How well conditioned is the SSIMULACRA2 score of a frozen case?  Moves a random fraction of the linear-RGB samples of both
frames by +-1 ulp (what any two non-bit-identical transfer functions do to each other) and reports the spread of the score.

    python tools/score_conditioning.py [kind w h pair matrix] [--frac 0.008] [--seeds 8]

Answers "how close to the accurate evaluation can an implementation that is not bit-identical with it get": the spread below
is the width of the target, and it is a property of the metric on that input (f32 cancellation in sigma - mu^2 against C2 at
the coarse scales, rectified by max(.,0)), not of this build.  Uses the C oracle (test infrastructure) as the evaluator."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_accurate as GA  # noqa: E402
from oracle import oracle as O  # noqa: E402


def perturb(lin, rng, frac):
    u = lin.view(np.int32).copy()
    m = (rng.random(u.shape) < frac) & (lin > 0) & (lin < 1)
    u[m] += rng.choice(np.array([-1, 1], np.int32), size=int(m.sum()))
    return u.view(np.float32)


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    case = (a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4])) if len(a) >= 5 else ("nv12", 1920, 1080, 2, 0)
    frac = float(sys.argv[sys.argv.index("--frac") + 1]) if "--frac" in sys.argv else 0.008
    seeds = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else 8
    lr, ld = GA.twin_linear_pair(*case, "exact")
    base = O.ssimulacra2_from_linear(lr, ld)[0]
    out = []
    for s in range(seeds):
        rng = np.random.default_rng(1000 + s)
        out.append(O.ssimulacra2_from_linear(perturb(lr, rng, frac), perturb(ld, rng, frac))[0] - base)
        print(f"seed {s}: {out[-1]:+.3e}", flush=True)
    out = np.array(out)
    print(json.dumps({"case": case, "frac": frac, "base": base, "delta_min": out.min(), "delta_max": out.max(), "delta_rms": float(np.sqrt((out ** 2).mean()))}))


if __name__ == "__main__":
    main()
