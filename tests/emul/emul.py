"""TEST INFRASTRUCTURE ONLY: drives tests/emul/libtm_emul.so, which executes the product's kernel
SOURCE (turbo-metrics_amd/csrc/tm_kernels.h) lane by lane on the CPU over host arenas laid out like
the engine's HBM arenas.  Lets the no-GPU tier check kernel indexing/ordering against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
# TM_EMUL_ASAN=1: AddressSanitizer build of the kernel source (GPU ASan is not available on the pool; every arena the kernels
# touch is a numpy allocation of exactly the engine's size, so out-of-bounds accesses of any kernel show up here).  Run with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 -- see tools/asan_emul.sh
_ASAN = os.environ.get("TM_EMUL_ASAN") == "1"
_LIB = os.path.join(_HERE, "libtm_emul_asan.so" if _ASAN else "libtm_emul.so")
_SRCS = [os.path.join(_HERE, "tm_emul.cpp"), os.path.join(_HERE, "hip_emul.h")] + [
    os.path.join(_ROOT, "turbo-metrics_amd", "csrc", f) for f in ("tm_platform.h", "tm_kernels.h", "tm_reference_kernels.h", "tm_ssim_kernels.h", "tm_device_math.h", "tm_geom.h", "tm_math_tables.inc")]


def build():
    if os.path.exists(_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(_LIB) for s in _SRCS):
        return _LIB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-fPIC"]
                          + (["-fsanitize=address", "-fno-omit-frame-pointer", "-g"] if _ASAN else [])
                          + ["-shared", "-pthread", "-Wno-unknown-pragmas", "-I", _HERE, "-o", _LIB, _SRCS[0]])
    return _LIB


def product_pow_tables():
    """the math table buffer of turbo-metrics_amd/csrc/tm_math_tables.inc as the kernels expect it (TM_TAB_DOUBLES doubles):
    96 doubles (rcp, nlog, exp2 of pow_pos), then the 513 x 4 binary64 coefficients of the BT.709 transfer-function cubics"""
    import re
    txt = open(os.path.join(_ROOT, "turbo-metrics_amd", "csrc", "tm_math_tables.inc")).read()
    dbl = [float.fromhex(v) for v in re.findall(r"(-?0x[0-9a-f.]+p[-+]?[0-9]+)", txt)]
    assert len(dbl) == 96 + 513 * 4, len(dbl)
    return np.array(dbl, np.float64)


class ScaleGeom(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pitch", C.c_int), ("pitch_t", C.c_int), ("plane", C.c_ulonglong),
                ("plane_t", C.c_ulonglong), ("off", C.c_ulonglong), ("off_t", C.c_ulonglong)]


class Geom(C.Structure):
    _fields_ = [("s", ScaleGeom * 6), ("pyr", C.c_ulonglong), ("pyr_t", C.c_ulonglong), ("vblk", C.c_int * 7), ("hblk", C.c_int * 7)]


class SsimGeom(C.Structure):
    _fields_ = [("w", C.c_int * 5), ("h", C.c_int * 5), ("pitch", C.c_int * 5), ("off", C.c_ulonglong * 5), ("qplane", C.c_ulonglong),
                ("pyr", C.c_ulonglong), ("strips_x", C.c_int * 5), ("segs_y", C.c_int * 5), ("seg_rows", C.c_int * 5), ("item_off", C.c_int * 6), ("g", C.c_float * 11)]


class FrameDesc(C.Structure):
    _fields_ = [("p0", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("pitch", C.c_ulonglong), ("pitch2", C.c_ulonglong),
                ("kind", C.c_int), ("matrix", C.c_int), ("shift", C.c_int), ("pad_", C.c_int)]


KIND = {"nv12": 0, "p016": 1, "rgb8": 2, "rgb16": 3, "rgbf32": 4, "linear_f32": 5, "i420_8": 6, "i420_16": 7, "i420_p10": 8}


class Emulated:
    """Runs one of the two pipelines (variant 0 = default, 1 = reference, 0x100 = default with the wide-frame row pass) for n slots; keeps the arenas for plane inspection."""

    def __init__(self, w, h, frames, lut, coef, want_sse=True, variant=0, powtab=None, weights=None, full_sums=True, ssim_window=None, ssim_need_l=31, ingest_rows=4):
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
                d = desc[2 * i + side]
                if f["kind"] == "i420":  # planar: data = (Y, Cb, Cr), uint8 or uint16 with the value in the low `bits` bits
                    pl = [np.ascontiguousarray(p) for p in f["data"]]; keep.extend(pl)
                    d.kind = KIND["i420_8" if f["bits"] == 8 else "i420_16"]; d.matrix = int(f.get("matrix", 0))
                    d.p0, d.p1, d.p2 = (p.ctypes.data for p in pl)
                    d.pitch, d.pitch2 = pl[0].strides[0], pl[1].strides[0]
                    d.shift = 0 if f["bits"] == 8 else 16 - f["bits"]
                    continue
                if f["kind"] == "i420p10":  # the packed 10-bit upload kind: data = (Y, Cb, Cr) as uint32 word planes (synth.p10_pack_plane)
                    pl = [np.ascontiguousarray(p, np.uint32) for p in f["data"]]; keep.extend(pl)
                    d.kind = KIND["i420_p10"]; d.matrix = int(f.get("matrix", 0))
                    d.p0, d.p1, d.p2 = (p.ctypes.data for p in pl)
                    d.pitch, d.pitch2 = pl[0].strides[0], pl[1].strides[0]
                    d.shift = 6
                    continue
                a = np.ascontiguousarray(f["data"]); keep.append(a)
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
        # SSIM / MS-SSIM (default pipeline only: its ingest kernel writes the quantised u8 planes): ssim_window = the 11 taps
        self.sg = None
        qu8, qplane, qpitch = None, 0, 0
        if ssim_window is not None:
            L.emul_ssim_geom_size.restype = C.c_size_t
            assert L.emul_ssim_geom_size() == C.sizeof(SsimGeom), (L.emul_ssim_geom_size(), C.sizeof(SsimGeom))
            gw = np.ascontiguousarray(ssim_window, np.float32)
            self.sg = SsimGeom()
            L.emul_ssim_geom(w, h, vp(gw), C.byref(self.sg))
            qplane, qpitch = self.sg.qplane, self.sg.pitch[0]
            self.QU8 = np.zeros(n * 2 * 3 * qplane, np.uint8)
            qu8 = vp(self.QU8)
        L.emul_set_ingest_rows(int(ingest_rows))
        L.emul_pipeline.argtypes = None
        L.emul_pipeline(w, h, n, desc, vp(lut), vp(coef), vp(powtab), int(want_sse), vp(self.LIN), vp(self.XYB), vp(self.XYBT), vp(self.V),
                        vp(self.PART), vp(self.SUMS), vp(self.SSE), int(variant), vp(weights), int(full_sums), qu8, C.c_ulonglong(qplane), int(qpitch))
        if ssim_window is not None:
            self.SPYR = np.zeros(max(1, n * 2 * 3 * self.sg.pyr), np.uint16)
            self.SPART = np.zeros(max(1, n * 3 * self.sg.item_off[5] * 2), np.float64)
            self.SSUMS = np.zeros(n * 30, np.float64)
            L.emul_ssim(w, h, n, vp(gw), qu8, vp(self.SPYR), vp(self.SPART), vp(self.SSUMS), int(ssim_need_l))
        self.SSE = self.SSE.reshape(n, -1).sum(axis=1)  # TM_SSE_BINS accumulators per slot
        self.w, self.h, self.n = w, h, n

    def plane(self, arena, slot, scale, index, channel, transposed=False, per_slot=2):
        sg = self.g.s[scale]
        if transposed:
            base = (slot * per_slot + index) * self.g.pyr_t + sg.off_t + channel * sg.plane_t
            return arena[base:base + sg.plane_t].reshape(-1, sg.pitch_t)[:sg.w, :sg.h]
        base = (slot * per_slot + index) * self.g.pyr + sg.off + channel * sg.plane
        return arena[base:base + sg.plane].reshape(sg.h, sg.pitch)[:, :sg.w]

    def qplane(self, slot, side, c):
        sg = self.sg
        base = ((slot * 2 + side) * 3 + c) * sg.qplane
        return self.QU8[base:base + sg.qplane].reshape(sg.h[0], sg.pitch[0])[:, :sg.w[0]]

    def ssim_sums(self, slot):
        return self.SSUMS[slot * 30:(slot + 1) * 30].reshape(3, 5, 2)

    def sums(self, slot):
        return self.SUMS[slot * 108:(slot + 1) * 108].reshape(6, 6, 3)
