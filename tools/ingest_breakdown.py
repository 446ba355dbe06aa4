#!/usr/bin/env python3
"""HISTORICAL (round 3): needs the -DTM_ABLATE_CBRT / -DTM_ABLATE_EOTF switches that lived in the kernel sources up to commit 3773fd2 (round 4);
the product sources carry no lab switches since round 5.  Its output is profiles/r03_ingest_rows_breakdown.txt.

Static cost breakdown of k_ingest_rows<NV12> (the SSIMULACRA2-only instantiation) BY PURPOSE, from ablation builds: the kernel is
compiled (device code only, nothing is run) as it ships and with one ingredient removed at a time (-DTM_ABLATE_*), and the
difference in instructions and in VALU pipe cycles of the loop body is that ingredient's cost.  VALU pipe cycles per 64-lane
instruction on gfx950: 2 for a full-rate f32 / integer instruction, 4 for a packed-f32 or binary64 one, 8 for v_mul_hi_u32 and
friends (MI355X_MICROARCH.md).  Output: profiles-ready text.

    python tools/ingest_breakdown.py > profiles/r03_ingest_rows_breakdown.txt
"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "turbo-metrics_amd", "csrc")
KEY = "_ZN3tmk13k_ingest_rowsILi0ELb0E"


def build(flags):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "tm_engine.hip")] + flags, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.remove(out)
    a = txt.index("\n" + KEY)
    body = txt[a:txt.index(".end_amdhsa_kernel", a)].split("\n")
    # the row loop = everything from the header of the largest loop to the end of the function (its latch is laid out before the
    # header, the epilogue after it is a handful of instructions); the two rare paths (linear branch of the transfer function,
    # zeroing of incomplete quads: wave-uniform branches marked "; rare path") are left out up to the next label
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    lo, hi = heads[-1], len(body) - 1
    keep, skipping = [], False
    for l in body[lo:hi + 1]:
        if "; rare path" in l:
            skipping = True
        elif l.startswith(".LBB"):
            skipping = False
        if not skipping:
            keep.append(l)
    body, lo, hi = keep, 0, len(keep) - 1
    st = {"instr": 0, "valu": 0, "cycles": 0, "salu": 0, "lds": 0, "vmem": 0, "packed": 0, "f64": 0}
    for l in body[lo:hi + 1]:
        t = l.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z_0-9]+$", op):
            continue
        st["instr"] += 1
        if op.startswith("v_"):
            st["valu"] += 1
            f64 = "_f64" in op
            pk = op.startswith("v_pk_")
            slow = op.split("_e")[0] in ("v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_mul_hi_i32")
            st["cycles"] += 8 if slow else (4 if (pk or f64) else 2)
            st["packed"] += pk
            st["f64"] += f64
        elif op.startswith("ds_"):
            st["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            st["vmem"] += 1
        elif op.startswith("s_"):
            st["salu"] += 1
    return st


def main():
    full = build([])
    print("k_ingest_rows<NV12, no quantised planes>: one iteration of the row loop = 64 quads (256 pixels) of BOTH frames -> 2 x 5 x 64 XYB pixels")
    print(f"  as shipped            : {full['instr']:5d} instructions, {full['valu']:5d} VALU ({full['packed']} packed f32, {full['f64']} binary64), "
          f"{full['salu']} SALU / wait / branch, {full['lds']} LDS, {full['vmem']} global; VALU pipe {full['cycles']} cycles")
    rows = [("transfer function (24 evaluations: f32 base, table index, binary64 cubic)", ["-DTM_ABLATE_EOTF"]),
            ("cube roots (30, as 15 pairs)", ["-DTM_ABLATE_CBRT"]),
            ("both", ["-DTM_ABLATE_EOTF", "-DTM_ABLATE_CBRT"])]
    for name, fl in rows:
        st = build(fl)
        print(f"  without {name:74s}: {st['instr']:5d} instructions, VALU pipe {st['cycles']:5d} cycles  ->  costs {full['instr'] - st['instr']:4d} instructions, {full['cycles'] - st['cycles']:5d} cycles "
              f"({100.0 * (full['cycles'] - st['cycles']) / full['cycles']:.0f} %)")
    both = build(["-DTM_ABLATE_EOTF", "-DTM_ABLATE_CBRT"])
    print(f"  the rest (sample unpack + YUV matrix, box filter, XYB mix and affine steps, level-2 pairing, addressing, stores): VALU pipe {both['cycles']} cycles "
          f"({100.0 * both['cycles'] / full['cycles']:.0f} %)")
    waves = 64 * 960 * 540 / 64  # wave-iterations per 64 1080p pairs
    print(f"  at 100 % VALU pipe use {full['cycles']} cycles x {waves:.0f} wave-iterations / (1024 SIMDs x 2.4 GHz) = {full['cycles'] * waves / 1024 / 2.4e9 * 1e3:.2f} ms per 64 1080p pairs")


if __name__ == "__main__":
    main()
