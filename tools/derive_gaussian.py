#!/usr/bin/env python3
"""Independent derivation of the sigma=1.5 truncated-cosine recursive Gaussian
(Charalampidis 2016) constants, to pin the literals used by the HIP kernels and
the oracle.  Follows the equations cited by the reference's generator
(crates/ssimulacra2-cuda-kernel/build.rs:28-145): (57) radius, Table I omega,
(37) p_k, (44) r_k, (50) rho_k, (52) zeta, (53)-(56) beta, (33) n2/d1.
Prints the f32 bit patterns; tests compare them with tests/golden/reference_tables.json.
"""
import math, struct
import numpy as np


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", np.float32(x)))[0]


def derive(sigma=1.5):
    radius = round(3.2795 * sigma + 0.2546)
    w = [k * math.pi / (2.0 * radius) for k in (1, 3, 5)]
    p = [1.0 / math.tan(0.5 * w[0]), -1.0 / math.tan(0.5 * w[1]), 1.0 / math.tan(0.5 * w[2])]
    r = [p[0] ** 2 / math.sin(w[0]), -p[1] ** 2 / math.sin(w[1]), p[2] ** 2 / math.sin(w[2])]
    rho = [math.exp(-0.5 * sigma * sigma * wk * wk) / radius for wk in w]
    d13 = p[0] * r[1] - r[0] * p[1]
    d35 = p[1] * r[2] - r[1] * p[2]
    d51 = p[2] * r[0] - r[2] * p[0]
    z15, z35 = d35 / d13, d51 / d13
    A = np.array([[p[0], p[1], p[2]], [r[0], r[1], r[2]], [z15, z35, 1.0]], dtype=np.float64)
    gamma = np.array([1.0, radius * radius - sigma * sigma, z15 * rho[0] + z35 * rho[1] + rho[2]])
    beta = np.linalg.solve(A, gamma)
    assert abs(beta[0] * p[0] + beta[1] * p[1] + beta[2] * p[2] - 1.0) < 1e-12
    n2 = [-beta[i] * math.cos(w[i] * (radius + 1.0)) for i in range(3)]
    d1 = [-2.0 * math.cos(w[i]) for i in range(3)]
    return radius, n2, d1


if __name__ == "__main__":
    radius, n2, d1 = derive()
    print("RADIUS", radius)
    for i, k in enumerate((1, 3, 5)):
        print(f"MUL_IN_{k}   = {np.float32(n2[i])!r:>16}  bits 0x{f32_bits(n2[i]):08x}")
        print(f"MUL_PREV_{k} = {np.float32(-d1[i])!r:>16}  bits 0x{f32_bits(-d1[i]):08x}")
