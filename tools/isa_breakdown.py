#!/usr/bin/env python3
"""Static instruction breakdown of one kernel of turbo-metrics_amd/csrc/tm_engine.gfx950.s (`make -C turbo-metrics_amd/csrc asm`).

    python tools/isa_breakdown.py k_ingest_wave 'ILi0E'      # substring(s) of the mangled name

Counts the instructions of the kernel body by issue class (packed f32, scalar-lane f32, conversions, integer / logic VALU,
quarter-rate VALU, LDS, global memory, SALU, branches, waits) and prints the register / LDS / occupancy figures the assembler
recorded.  Static counts: a loop body counts once (the two-sided loop of k_ingest_wave executes twice per wave).
"""
import collections
import os
import re
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32",
           "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


def classify(op):
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "valu packed f32 (2 lanes-worth per issue)"
    if op in QUARTER or op.endswith("_f64") and op.startswith("v_") and not op.startswith("v_cvt"):
        return "valu quarter / f64 rate"
    if op.startswith("v_cvt") or op.startswith("v_rndne") or op.startswith("v_fract") or op.startswith("v_floor") or op.startswith("v_trunc"):
        return "valu convert / round"
    if op.startswith("v_") and (op.endswith("_f32") or op.endswith("_f32_e32") or op.endswith("_f32_e64") or "_f32_" in op):
        return "valu f32"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "valu compare / select"
    if op.startswith("v_"):
        return "valu integer / logic / move"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "global load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store") or op.startswith("scratch_store"):
        return "global store"
    if op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "global atomic"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait / barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar load"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_bodies(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end") or line.lstrip().startswith(".end_amdhsa_kernel"):
            pass
        if line.startswith("\t.section") or line.startswith("\t.amdhsa_kernel"):
            if body:
                yield name, body
            name, body = None, []
            continue
        body.append(line)


def meta(path, name):
    out, on = {}, False
    for line in open(path):
        if line.startswith("\t.amdhsa_kernel " + name):
            on = True
        elif on and ".end_amdhsa_kernel" in line:
            break
        elif on:
            m = re.match(r"\s+\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|accum_offset|private_segment_fixed_size)\s+(\d+)", line)
            if m:
                out[m.group(1)] = int(m.group(2))
    return out


def main():
    path = os.path.join(ROOT, "turbo-metrics_amd", "csrc", "tm_engine.gfx950.s")
    keys = sys.argv[1:] or ["k_ingest_wave", "ILi0E"]
    for name, body in kernel_bodies(path):
        if not all(k in name for k in keys):
            continue
        cnt = collections.Counter()
        ops = collections.Counter()
        for line in body:
            t = line.strip()
            if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if not re.match(r"^[a-z_0-9]+$", op):
                continue
            cnt[classify(op)] += 1
            ops[op] += 1
        total = sum(cnt.values())
        valu = sum(v for k, v in cnt.items() if k.startswith("valu"))
        print(f"== {name}\n   {total} instructions, {valu} VALU; {meta(path, name)}")
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
            print(f"   {v:6d}  {k}")
        print("   top opcodes: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(24)))


if __name__ == "__main__":
    main()
