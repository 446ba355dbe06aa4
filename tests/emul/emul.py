"""TEST INFRASTRUCTURE ONLY: drives tests/emul/libtm_emul.so, which executes the product's kernel
SOURCE (turbo-metrics_amd/csrc/tm_kernels.h) lane by lane on the CPU over host arenas laid out like
the engine's HBM arenas.  Lets the no-GPU tier check kernel indexing/ordering against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = os.path.join(_HERE, "libtm_emul.so")
_SRCS = [os.path.join(_HERE, "tm_emul.cpp"), os.path.join(_HERE, "hip_emul.h")] + [
    os.path.join(_ROOT, "turbo-metrics_amd", "csrc", f) for f in ("tm_kernels.h", "tm_device_math.h", "tm_geom.h", "tm_math_tables.inc")]


def build():
    if os.path.exists(_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(_LIB) for s in _SRCS):
        return _LIB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                           "-shared", "-pthread", "-Wno-unknown-pragmas", "-I", _HERE, "-o", _LIB, _SRCS[0]])
    return _LIB


def product_pow_tables():
    """the 96 doubles of turbo-metrics_amd/csrc/tm_math_tables.inc (rcp, nlog, exp2)"""
    import re
    txt = open(os.path.join(_ROOT, "turbo-metrics_amd", "csrc", "tm_math_tables.inc")).read()
    vals = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[-+]?[0-9]+", txt)]
    assert len(vals) == 96
    return np.array(vals, np.float64)


class ScaleGeom(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pitch", C.c_int), ("pitch_t", C.c_int), ("plane", C.c_ulonglong),
                ("plane_t", C.c_ulonglong), ("off", C.c_ulonglong), ("off_t", C.c_ulonglong)]


class Geom(C.Structure):
    _fields_ = [("s", ScaleGeom * 6), ("pyr", C.c_ulonglong), ("pyr_t", C.c_ulonglong), ("vblk", C.c_int * 7), ("hblk", C.c_int * 7)]


class FrameDesc(C.Structure):
    _fields_ = [("p0", C.c_void_p), ("p1", C.c_void_p), ("pitch", C.c_ulonglong), ("kind", C.c_int), ("matrix", C.c_int)]


KIND = {"nv12": 0, "p016": 1, "rgb8": 2, "rgb16": 3, "rgbf32": 4, "linear_f32": 5}


class Emulated:
    """Runs the whole generation-0 pipeline for n slots; keeps the arenas for plane inspection."""

    def __init__(self, w, h, frames, lut, coef, want_sse=True, variant=0, powtab=None, weights=None, full_sums=True):
        """frames: list of (ref, dis) where each is dict(kind=, data=np.ndarray, pitch=, coded_height=, matrix=)."""
        L = C.CDLL(build())
        L.emul_geom_size.restype = C.c_size_t
        assert L.emul_geom_size() == C.sizeof(Geom), (L.emul_geom_size(), C.sizeof(Geom))
        self.g = Geom()
        L.emul_geom(w, h, C.byref(self.g))
        n = len(frames)
        sizes = (C.c_ulonglong * 7)()
        L.emul_sizes(w, h, n, sizes)
        self.LIN = np.zeros(sizes[0], np.float32); self.XYB = np.zeros(sizes[1], np.float32)
        self.XYBT = np.zeros(sizes[2], np.float32); self.V = np.zeros(sizes[3], np.float32)
        self.PART = np.zeros(sizes[4], np.float64); self.SUMS = np.zeros(sizes[5], np.float64)
        self.SSE = np.zeros(sizes[6], np.uint64)
        desc = (FrameDesc * (2 * n))()
        keep = []
        for i, pair in enumerate(frames):
            for side, f in enumerate(pair):
                a = np.ascontiguousarray(f["data"]); keep.append(a)
                d = desc[2 * i + side]
                d.kind = KIND[f["kind"]]; d.matrix = int(f.get("matrix", 0)); d.p0 = a.ctypes.data
                if f["kind"] in ("nv12", "p016"):
                    d.pitch = f["pitch"]; d.p1 = a.ctypes.data + f["pitch"] * f["coded_height"]
                else:
                    d.pitch = a.strides[0]; d.p1 = None
        lut = np.ascontiguousarray(lut, np.float32); coef = np.ascontiguousarray(coef, np.float32)
        powtab = np.ascontiguousarray(product_pow_tables() if powtab is None else powtab, np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        if weights is None:  # [channel][scale][6]; all non-zero -> every job FULL
            weights = np.ones(108)
        weights = np.ascontiguousarray(weights, np.float64).ravel()
        assert weights.size == 108
        L.emul_pipeline(w, h, n, desc, vp(lut), vp(coef), vp(powtab), int(want_sse), vp(self.LIN), vp(self.XYB), vp(self.XYBT), vp(self.V),
                        vp(self.PART), vp(self.SUMS), vp(self.SSE), int(variant), vp(weights), int(full_sums))
        self.w, self.h, self.n = w, h, n

    def plane(self, arena, slot, scale, index, channel, transposed=False, per_slot=2):
        sg = self.g.s[scale]
        if transposed:
            base = (slot * per_slot + index) * self.g.pyr_t + sg.off_t + channel * sg.plane_t
            return arena[base:base + sg.plane_t].reshape(-1, sg.pitch_t)[:sg.w, :sg.h]
        base = (slot * per_slot + index) * self.g.pyr + sg.off + channel * sg.plane
        return arena[base:base + sg.plane].reshape(sg.h, sg.pitch)[:, :sg.w]

    def sums(self, slot):
        return self.SUMS[slot * 108:(slot + 1) * 108].reshape(6, 6, 3)
