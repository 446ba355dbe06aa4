#!/usr/bin/env python3
"""Extract the reference's known-answer DATA tables into golden fixtures.

Runs only in the build container (reads /root/reference, which never travels to
the GPU box).  Outputs are pure data (numbers), no reference source text:

  tests/golden/reference_tables.json
      srgb_lut_bits  : 256 u32 bit patterns of SRGB8_TO_LINEARF32_LUT
                       (crates/cuda-colorspace-kernel/src/srgb.rs:5-38; the same table
                        is repeated in ssimulacra2-cuda-kernel/src/srgb.rs:18-51 and
                        ssimulacra2-cuda/examples/cpu.rs:20-277 -- all three are
                        checked to be identical here)
      weights        : 108 f64 (crates/ssimulacra2-cuda/src/lib.rs:454-584, checked equal
                       to examples/cpu.rs:729-838)
      gaussian       : the recursive-gaussian f32 literals of examples/cpu.rs:931-948
"""
import json, re, struct, sys, os

REF = "/root/reference/crates"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_tables.json")


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def num_list(body):
    body = re.sub(r"//[^\n]*", "", body)
    toks = re.findall(r"[-+]?[0-9][0-9_]*\.?[0-9_]*(?:[eE][-+]?[0-9]+)?", body)
    return [float(t.replace("_", "")) for t in toks]


def lut_from(path, name):
    src = open(path).read()
    m = re.search(name + r": \[f32; 256\] = \[(.*?)\];", src, re.S)
    v = num_list(m.group(1))
    assert len(v) == 256, (path, len(v))
    return [f32_bits(x) for x in v]


def weights_from(path):
    src = open(path).read()
    m = re.search(r"const WEIGHT: \[f64; 108\] = \[(.*?)\];", src, re.S)
    v = num_list(m.group(1))
    assert len(v) == 108, (path, len(v))
    return v


def main():
    a = lut_from(f"{REF}/cuda-colorspace-kernel/src/srgb.rs", "SRGB8_TO_LINEARF32_LUT")
    b = lut_from(f"{REF}/ssimulacra2-cuda-kernel/src/srgb.rs", "SRGB8_TO_LINEARF32_LUT")
    c = lut_from(f"{REF}/ssimulacra2-cuda/examples/cpu.rs", "FROM_SRGB8_TABLE")
    assert a == b == c, "the three LUT copies differ"
    w1 = weights_from(f"{REF}/ssimulacra2-cuda/src/lib.rs")
    w2 = weights_from(f"{REF}/ssimulacra2-cuda/examples/cpu.rs")
    assert w1 == w2, "weight tables differ"
    src = open(f"{REF}/ssimulacra2-cuda/examples/cpu.rs").read()
    g = {}
    for name, val in re.findall(r"pub const (\w+): (?:f32|usize) = ([-0-9.e_]+?)_(?:f32|usize);", src):
        g[name] = float(val.replace("_", "")) if name != "RADIUS" else int(val)
    out = {
        "srgb_lut_bits": a,
        "weights": [repr(x) for x in w1],
        "gaussian": {k: (v if k == "RADIUS" else f32_bits(v)) for k, v in g.items()},
        "known_answers": {
            "identical_inputs_score": 100.0,
            "npp_sum_128x128_r1": [16384.0, 0.0, 0.0],
        },
    }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, "nonzero weights:", sum(1 for x in w1 if x != 0.0))


if __name__ == "__main__":
    main()
