#!/usr/bin/env python3
"""GPU box: does the distance between the planes / slots of the pass-1 arena (g.pyr_t floats; slots are 5 x that apart) decide how
fast the column pass writes it?  For each padding (TM_PYRT_PAD floats, multiples of 64) a few fresh engines without the placement
search; column-pass and row-pass ms per 64 1080p pairs of each.  One process per padding (the variable is read at creation)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch
from tm_pkg import tm
w, h, B = 1920, 1080, 64
tm.init_hip(0)
tm.set_placement_candidates(1)
pairs = []
for n in range(2):
    (rs, rp, rch), (ds, dp, dch) = tm.synth.nv12_pair(w, h, n)
    pairs.append(((torch.from_numpy(rs).cuda(), rp, rch), (torch.from_numpy(ds).cuda(), dp, dch)))
out = []
engs = []
for k in range(int(sys.argv[1])):
    eng = tm.TurboMetrics(w, h, tm.Metrics(ssimulacra2=True), batch=B)
    engs.append(eng)  # keep them alive: every engine lands somewhere else
    for slot in range(B):
        (rt, rp, rch), (dt, dp, dch) = pairs[slot %% 2]
        eng.set_pair(slot, tm.HwFrame.nv12(rt, rp, rch), tm.HwFrame.nv12(dt, dp, dch))
    eng.set_profiling(True)
    for _ in range(25):
        eng.compute_async(); eng.sync()
    eng.stage_ms(reset=True)
    for _ in range(20):
        eng.compute_async(); eng.sync()
    ms, n = eng.stage_ms(reset=True)
    out.append([round(ms[1] / n, 3), round(ms[2] / n, 3)])
print(json.dumps(out))
''' % ROOT
pads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,64,1024,4160,65600,1048640".split(","))]
for pad in pads:
    env = dict(os.environ, TM_PYRT_PAD=str(pad))
    r = subprocess.run([sys.executable, "-c", CHILD, "4"], capture_output=True, text=True, env=env, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("[")]
    print(json.dumps({"pyr_t_pad_floats": pad, "col_row_ms_per_engine": json.loads(line[0]) if line else r.stderr[-300:]}), flush=True)
