"""Host-side mirror of the reference's operator interface for the frame-pair path, over the C ABI.

Reference types mirrored (paths relative to /root/reference/crates):
  Metrics, Options, FrameScores      turbo-metrics/src/lib.rs:27-54,112-123
  HwFrame                            turbo-metrics/src/lib.rs:125-130
  TurboMetrics::{new,compute_one,compute_all}   turbo-metrics/src/lib.rs:201-433
  Ssimulacra2::{new,compute_sync,mem_usage}     ssimulacra2-cuda/src/lib.rs:48-138,271-291
  ColorMatrix                        cuda-colorspace/src/lib.rs
  init_cuda                          turbo-metrics/src/lib.rs:438-456  (here: init_hip)

The reference is Rust; this image has no Rust toolchain, so the mirror is Python (tests and the
benchmark read like the reference's call sites).  All arithmetic happens in libturbometrics_hip.so.
"""
import ctypes as C
import enum
from dataclasses import dataclass
from typing import Iterable, List, Optional

import numpy as np

from . import ffi


class TmError(RuntimeError):
    def __init__(self, code, where):
        L = ffi.lib()
        msg = L.tm_strerror(code).decode()
        hip = L.tm_last_hip_error().decode()
        super().__init__(f"{where}: {msg}" + (f" [{hip}]" if code == ffi.TM_ERR_HIP and hip else ""))
        self.code = code


def _chk(code, where):
    if code != ffi.TM_OK:
        raise TmError(code, where)


def init_hip(device: int = 0):
    """Counterpart of init_cuda(): bind this process to `device`; raises when no gfx950 GPU exists."""
    _chk(ffi.lib().tm_init(int(device)), "tm_init")


def set_debug_log(on: bool):
    """tm_set_debug_log: diagnostics on stderr (what the placement search measured per candidate)"""
    ffi.lib().tm_set_debug_log(int(bool(on)))


def set_placement_candidates(n: int):
    """tm_set_placement_candidates: allocations of the pass-1 arena that engine creation tries (it keeps the one on which the
    column pass runs fastest); 1 = off.  Process-wide, applies to engines created afterwards."""
    ffi.lib().tm_set_placement_candidates(int(n))


class ColorMatrix(enum.IntEnum):
    BT709 = ffi.TM_MATRIX_BT709
    BT601_525 = ffi.TM_MATRIX_BT601_525
    BT601_625 = ffi.TM_MATRIX_BT601_625


def color_matrix_fallback(height: int) -> ColorMatrix:
    """turbo-metrics/src/color.rs:51-78: unspecified metadata falls back by frame height."""
    if height <= 525:
        return ColorMatrix.BT601_525
    if height <= 625:
        return ColorMatrix.BT601_625
    return ColorMatrix.BT709


@dataclass
class Metrics:
    psnr: bool = False
    ssim: bool = False
    msssim: bool = False
    ssimulacra2: bool = False

    def mask(self) -> int:
        return ((ffi.TM_METRIC_PSNR if self.psnr else 0) | (ffi.TM_METRIC_SSIM if self.ssim else 0)
                | (ffi.TM_METRIC_MSSSIM if self.msssim else 0) | (ffi.TM_METRIC_SSIMULACRA2 if self.ssimulacra2 else 0))


@dataclass
class Options:
    every: int = 0
    skip: int = 0
    skip_ref: int = 0
    skip_dis: int = 0
    frames: int = 0


@dataclass
class FrameScores:
    psnr: Optional[float] = None
    ssim: Optional[float] = None
    msssim: Optional[float] = None
    ssimulacra2: Optional[float] = None


@dataclass
class HwFrame:
    """One decoded frame.  kind: 'nv12' | 'p016' | 'i420' | 'i420p10' | 'rgb8' | 'rgb16' | 'rgbf32' | 'linear_f32'.
    'i420': planar 4:2:0, `data` = (Y, Cb, Cr) arrays (uint8, or uint16 with the value in the low `bits` bits).
    'i420p10': the same 10-bit planes PACKED three samples to a uint32 (tm_engine_set_frame_i420p10: the upload form; synth.p10_pack_plane).
    data: numpy array (host memory) or any object with .data_ptr() (device memory, e.g. a torch
    tensor on the GPU).  For the biplanar kinds `data` is the whole surface (luma rows, then the CbCr
    plane at pitch*coded_height); for RGB kinds it is (h, w, 3).
    Lifetime and mutation: a numpy array or a pageable CPU tensor is copied before set_frame returns.  A device tensor is
    BORROWED and a page-locked (pinned) CPU tensor is read by an asynchronous DMA: neither may be modified or freed until
    TurboMetrics.sync() has returned for the batch it was set for -- a loader that refills one pinned staging tensor per frame
    must wait for sync() (or use one staging tensor per slot in flight, as bench.py's host-fed leg and the CLI's frame ring do)."""
    kind: str
    data: object
    pitch: int = 0
    coded_height: int = 0
    matrix: ColorMatrix = ColorMatrix.BT709
    full_range: bool = False
    transfer: int = ffi.TM_TRANSFER_BT709
    bits: int = 8

    @staticmethod
    def i420(y, u, v, bits=8, matrix=ColorMatrix.BT709, full_range=False):
        return HwFrame("i420", (y, u, v), 0, 0, matrix, full_range, bits=bits)

    @staticmethod
    def i420p10(y, u, v, matrix=ColorMatrix.BT709, full_range=False):
        return HwFrame("i420p10", (y, u, v), 0, 0, matrix, full_range, bits=10)

    @staticmethod
    def nv12(surface, pitch, coded_height, matrix=ColorMatrix.BT709, full_range=False):
        return HwFrame("nv12", surface, pitch, coded_height, matrix, full_range)

    @staticmethod
    def p016(surface, pitch, coded_height, matrix=ColorMatrix.BT709, full_range=False):
        return HwFrame("p016", surface, pitch, coded_height, matrix, full_range)

    @staticmethod
    def rgb(arr):
        a = np.asarray(arr) if not hasattr(arr, "data_ptr") else arr
        dt = str(a.dtype).replace("torch.", "")
        kind = {"uint8": "rgb8", "uint16": "rgb16", "float32": "rgbf32"}[dt]
        return HwFrame(kind, a)

    @staticmethod
    def linear(arr):
        return HwFrame("linear_f32", arr)


def _ptr_and_mem(data):
    if hasattr(data, "data_ptr"):  # torch tensor
        mem = ffi.TM_MEM_DEVICE if getattr(data, "is_cuda", False) else (ffi.TM_MEM_HOST_PINNED if data.is_pinned() else ffi.TM_MEM_HOST)
        return int(data.data_ptr()), mem, data
    a = np.ascontiguousarray(data)
    return a.ctypes.data, ffi.TM_MEM_HOST, a


class TurboMetrics:
    """Mirror of turbo_metrics::TurboMetrics with `batch` frame-pair slots (batch=1 == reference)."""

    def __init__(self, width: int, height: int, metrics: Metrics, batch: int = 1, full_sums: bool = False):
        """full_sums: also compute the 56 of the 108 per-scale sums whose weight in the reference's table is 0.0
        (raw_sums() then equals the reference's `scores` array entry for entry; the score is the same either way)."""
        self._L = ffi.lib()
        self.width, self.height, self.batch = int(width), int(height), int(batch)
        self._metrics = metrics
        h = C.c_void_p()
        _chk(self._L.tm_engine_create(C.byref(h), self.width, self.height, metrics.mask(), self.batch), "tm_engine_create")
        self._h = h
        self._keep = {}
        if full_sums:
            self.set_full_sums(True)

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        d = getattr(self, "_def", None)
        if d is not None:
            self._def = None
            for p in d["peers"]:
                p.close()
        if getattr(self, "_h", None):
            self._L.tm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def metrics(self) -> Metrics:
        return self._metrics

    def mem_usage(self) -> int:
        return int(self._L.tm_engine_mem_usage(self._h))

    # -- frames -------------------------------------------------------------------------------
    def set_frame(self, slot: int, side: int, f: HwFrame):
        if f.kind in ("i420", "i420p10"):
            planes = [_ptr_and_mem(p) for p in f.data]
            self._keep[(slot, side)] = [k for _, _, k in planes]
            mems = {m for _, m, _ in planes}
            if len(mems) != 1:
                raise ValueError("the three planes must live in the same kind of memory")
            pitch = lambda k: int(k.stride(0) * k.element_size()) if hasattr(k, "data_ptr") else int(k.strides[0])
            if pitch(planes[1][2]) != pitch(planes[2][2]):
                raise ValueError("Cb and Cr must have the same row pitch (tm_engine_set_frame_i420 takes pitch_uv once)")
            if f.kind == "i420p10":
                _chk(self._L.tm_engine_set_frame_i420p10(self._h, slot, side, planes[0][0], planes[1][0], planes[2][0], pitch(planes[0][2]),
                                                         pitch(planes[1][2]), int(f.matrix), int(f.transfer),
                                                         int(bool(f.full_range)), mems.pop()), "tm_engine_set_frame_i420p10")
                return
            _chk(self._L.tm_engine_set_frame_i420(self._h, slot, side, planes[0][0], planes[1][0], planes[2][0], pitch(planes[0][2]),
                                                  pitch(planes[1][2]), int(f.bits), int(f.matrix), int(f.transfer),
                                                  int(bool(f.full_range)), mems.pop()), "tm_engine_set_frame_i420")
            return
        ptr, mem, keep = _ptr_and_mem(f.data)
        self._keep[(slot, side)] = keep  # device pointers must outlive the compute
        L, h = self._L, self._h
        if f.kind in ("nv12", "p016"):
            # a decoder surface: ONE allocation, coded_height luma rows, then the CbCr rows (NvDecNV12 / NvDecP016::from_mapping)
            fn = L.tm_engine_set_surface_nv12 if f.kind == "nv12" else L.tm_engine_set_surface_p016
            _chk(fn(h, slot, side, ptr, f.pitch, int(f.coded_height), int(f.matrix), int(f.transfer), int(bool(f.full_range)), mem),
                 f"tm_engine_set_surface_{f.kind}")
        else:
            shape = tuple(f.data.shape)
            if len(shape) != 3 or shape[2] != 3 or shape[0] != self.height or shape[1] != self.width:
                raise ValueError(f"expected ({self.height}, {self.width}, 3), got {shape}")
            bps = {"rgb8": 1, "rgb16": 2, "rgbf32": 4, "linear_f32": 4}[f.kind]
            fn = getattr(L, "tm_engine_set_frame_" + f.kind)
            _chk(fn(h, slot, side, ptr, self.width * 3 * bps, mem), f"tm_engine_set_frame_{f.kind}")

    def set_pair(self, slot: int, fref: HwFrame, fdis: HwFrame):
        self.set_frame(slot, ffi.TM_SIDE_REF, fref)
        self.set_frame(slot, ffi.TM_SIDE_DIS, fdis)

    # -- compute ------------------------------------------------------------------------------
    def compute_async(self, n_slots: Optional[int] = None):
        _chk(self._L.tm_engine_compute_async(self._h, self.batch if n_slots is None else int(n_slots)), "tm_engine_compute_async")

    def sync(self):
        _chk(self._L.tm_engine_sync(self._h), "tm_engine_sync")

    def upload_fence(self) -> int:
        """token for "every frame upload enqueued on this engine so far" (page-locked host frames: see the header)"""
        t = C.c_uint64()
        _chk(self._L.tm_engine_upload_fence(self._h, C.byref(t)), "tm_engine_upload_fence")
        return int(t.value)

    def upload_done(self, token: int, block: bool = False) -> bool:
        """have the uploads in front of `token` left host memory?  block=True waits for them"""
        r = self._L.tm_engine_upload_done(self._h, int(token), int(bool(block)))
        _chk(-r if r < 0 else ffi.TM_OK, "tm_engine_upload_done")
        return r > 0

    def scores(self, slot: int) -> FrameScores:
        s = ffi.FrameScoresC()
        _chk(self._L.tm_engine_get_scores(self._h, slot, C.byref(s)), "tm_engine_get_scores")
        v = s.valid
        return FrameScores(
            psnr=s.psnr if v & ffi.TM_METRIC_PSNR else None, ssim=s.ssim if v & ffi.TM_METRIC_SSIM else None,
            msssim=s.msssim if v & ffi.TM_METRIC_MSSSIM else None,
            ssimulacra2=s.ssimulacra2 if v & ffi.TM_METRIC_SSIMULACRA2 else None)

    def scores_batch(self, n: Optional[int] = None, first: int = 0) -> List[FrameScores]:
        """FrameScores of slots [first, first + n) of the last completed compute, one call across the ABI"""
        n = self.batch - first if n is None else int(n)
        arr = (ffi.FrameScoresC * n)()
        _chk(self._L.tm_engine_get_scores_batch(self._h, first, n, arr), "tm_engine_get_scores_batch")
        out = []
        for s in arr:
            v = s.valid
            out.append(FrameScores(
                psnr=s.psnr if v & ffi.TM_METRIC_PSNR else None, ssim=s.ssim if v & ffi.TM_METRIC_SSIM else None,
                msssim=s.msssim if v & ffi.TM_METRIC_MSSSIM else None,
                ssimulacra2=s.ssimulacra2 if v & ffi.TM_METRIC_SSIMULACRA2 else None))
        return out

    def raw_sums(self, slot: int) -> np.ndarray:
        out = np.zeros(108, np.float64)
        _chk(self._L.tm_engine_get_raw_sums(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_double))), "tm_engine_get_raw_sums")
        return out.reshape(6, 6, 3)

    def ssim_sums(self, slot: int) -> np.ndarray:
        """(3 channels, 5 scales, [sum ssim, sum cs]) of the quantised pair (build-defined SSIM / MS-SSIM, see the header)"""
        out = np.zeros(30, np.float64)
        _chk(self._L.tm_engine_get_ssim_sums(self._h, slot, out.ctypes.data_as(C.POINTER(C.c_double))), "tm_engine_get_ssim_sums")
        return out.reshape(3, 5, 2)

    def sse(self, slot: int) -> int:
        v = C.c_uint64()
        _chk(self._L.tm_engine_get_sse(self._h, slot, C.byref(v)), "tm_engine_get_sse")
        return int(v.value)

    def sse_channels(self, slot: int):
        v = (C.c_uint64 * 3)()
        _chk(self._L.tm_engine_get_sse_channels(self._h, slot, v), "tm_engine_get_sse_channels")
        return [int(x) for x in v]

    def _peer_follows(self, name, *args):
        """settings are the ENGINE's, and compute_one_deferred runs the pairs on further engines in turn: they follow (those set before
        they existed are replayed when they are created)"""
        self._settings = getattr(self, "_settings", {})
        self._settings[name] = args
        d = getattr(self, "_def", None)
        if d is not None:
            for p in d["peers"]:
                getattr(p, name)(*args)

    def set_channel_mode(self, first_channel_only: bool):
        """PSNR / SSIM / MS-SSIM from channel 0 only instead of pooled / averaged over R, G, B (see the header)"""
        self._retire_deferred()  # a pair in flight is scored with the mode it was submitted under
        _chk(self._L.tm_engine_set_channel_mode(self._h, ffi.TM_CHANNELS_FIRST if first_channel_only else ffi.TM_CHANNELS_POOLED), "tm_engine_set_channel_mode")
        self._peer_follows("set_channel_mode", first_channel_only)

    def _retire_deferred(self):
        """finish the pairs in flight for compute_one_deferred and keep their scores for collect()"""
        d = getattr(self, "_def", None)
        if d is not None:
            for i, e in enumerate([self] + d["peers"]):  # (no reference to self inside _def: an engine dropped without close() must
                if d["pending"][i] is not None:          # not wait for the cyclic collector with several engines' worth of HBM)
                    e.sync()
                    d["done"][d["pending"][i]] = e.scores(0)
                    d["pending"][i] = None

    MAX_DEFERRED_DEPTH = 8

    def set_deferred_depth(self, depth: int, create_now: bool = False):
        """pairs in flight at most for compute_one_deferred (default 2; == TurboMetrics::set_deferred_depth, host/turbo_metrics.hpp):
        one pair leaves most of the chip idle, three / four in flight reach 6.8 k / 8.3 k pairs/s of 1080p with frames in HBM -- for a
        caller that collects pair k after submitting pair k + depth - 1.  Pairs in flight are finished first (their scores stay
        collectable); engines beyond the new depth are freed; create_now: the engines of the turn are created by this call
        instead of when their turn first comes."""
        if not 2 <= int(depth) <= self.MAX_DEFERRED_DEPTH:
            raise TmError(ffi.TM_ERR_INVALID_ARG, "set_deferred_depth: 2 ... 8 pairs in flight")
        self._retire_deferred()
        self._def_depth = int(depth)
        d = getattr(self, "_def", None)
        if d is not None:
            for p in d["peers"][self._def_depth - 1:]:
                p.close()
            del d["peers"][self._def_depth - 1:]
            d["pending"] = [None] * self._def_depth
        if create_now:
            if self.batch != 1:
                raise ValueError("compute_one_deferred is the one-pair-per-call path: create the engine with batch=1")
            if getattr(self, "_def", None) is None:
                self._def = {"peers": [], "pending": [None] * self._def_depth, "done": {}, "next": 0}
            self._grow_peers(self._def_depth - 1)

    def _grow_peers(self, n: int):
        d = self._def
        while len(d["peers"]) < n:
            peer = TurboMetrics(self.width, self.height, self._metrics, batch=1)
            for name, args in getattr(self, "_settings", {}).items():  # channel mode, full sums, variant, graph: as set on this engine
                getattr(peer, name)(*args)
            d["peers"].append(peer)

    def compute_one(self, fref: HwFrame, fdis: HwFrame) -> FrameScores:
        """== TurboMetrics::compute_one: convert, compute, block, return FrameScores."""
        self._retire_deferred()  # (slot 0 may hold a deferred pair)
        self.set_pair(0, fref, fdis)
        self.compute_async(1)
        self.sync()
        return self.scores(0)

    def compute_one_deferred(self, fref: HwFrame, fdis: HwFrame) -> int:
        """compute_one without its blocking stream sync (lib.rs:352): hand the pair over, launch, return a ticket at once;
        collect(ticket) blocks until THAT pair's scores are there.  Two launches may be in flight (set_deferred_depth: up to eight) --
        as many engines taking turns, each created when its turn first comes (host/turbo_metrics.hpp: the same methods on the C++
        side).  One submission more first finishes the oldest pair and keeps its scores until collected.  Scores are bit-identical
        with compute_one's."""
        if self.batch != 1:
            raise ValueError("compute_one_deferred is the one-pair-per-call path: create the engine with batch=1")
        depth = getattr(self, "_def_depth", 2)
        if getattr(self, "_def", None) is None:
            self._def = {"peers": [], "pending": [None] * depth, "done": {}, "next": 0}
        d = self._def
        ticket = d["next"]
        d["next"] += 1
        i = ticket % depth
        self._grow_peers(i)  # the engines the launches take turns on are created when their turn first comes (or by set_deferred_depth)
        e = self if i == 0 else d["peers"][i - 1]
        if d["pending"][i] is not None:
            e.sync()
            d["done"][d["pending"][i]] = e.scores(0)
        e.set_pair(0, fref, fdis)
        e.compute_async(1)
        d["pending"][i] = ticket
        return ticket

    def collect(self, ticket: int) -> FrameScores:
        d = getattr(self, "_def", None)
        if d is not None:
            for i in range(len(d["pending"])):
                if d["pending"][i] == ticket:
                    e = self if i == 0 else d["peers"][i - 1]
                    d["pending"][i] = None
                    e.sync()
                    return e.scores(0)
            if ticket in d["done"]:
                return d["done"].pop(ticket)
        raise TmError(ffi.TM_ERR_INVALID_ARG, "collect: no such ticket (never issued, or collected already)")

    def compute_all(self, frames_ref: Iterable[HwFrame], frames_dis: Iterable[HwFrame], opts: Options = Options()) -> List[FrameScores]:
        """== TurboMetrics::compute_all frame selection (lib.rs:385-404), batched over the slots."""
        self._retire_deferred()
        it_r, it_d = iter(frames_ref), iter(frames_dis)
        for _ in range(opts.skip_ref + opts.skip):
            next(it_r, None)
        for _ in range(opts.skip_dis + opts.skip):
            next(it_d, None)
        out: List[FrameScores] = []
        decode_count, filled = 0, 0

        def flush():
            nonlocal filled
            if filled:
                self.compute_async(filled)
                self.sync()
                out.extend(self.scores(i) for i in range(filled))
                filled = 0

        for fr, fd in zip(it_r, it_d):
            if opts.every > 1 and decode_count != 0 and decode_count % opts.every != 0:
                decode_count += 1
                continue
            if opts.frames > 0 and decode_count >= opts.frames:
                break
            decode_count += 1
            self.set_pair(filled, fr, fd)
            filled += 1
            if filled == self.batch:
                flush()
        flush()
        return out

    # -- measurement / test hooks -------------------------------------------------------------
    def set_profiling(self, on: bool):
        _chk(self._L.tm_engine_set_profiling(self._h, int(on)), "tm_engine_set_profiling")

    def stage_ms(self, reset: bool = False):
        ms = (C.c_double * ffi.TM_STAGE_COUNT)()
        n = C.c_uint64()
        _chk(self._L.tm_engine_get_stage_ms(self._h, ms, C.byref(n), int(reset)), "tm_engine_get_stage_ms")
        return list(ms), int(n.value)

    def set_full_sums(self, on: bool):
        self._retire_deferred()  # tm_engine_set_full_sums drops the engine's results: collect the pair in flight first
        _chk(self._L.tm_engine_set_full_sums(self._h, int(bool(on))), "tm_engine_set_full_sums")
        self._peer_follows("set_full_sums", on)

    def job_modes(self) -> np.ndarray:
        """(6 scales, 3 channels): 0 = nothing computed, 1 = edge terms only, 2 = all three error maps"""
        out = (C.c_int * 18)()
        _chk(self._L.tm_engine_get_job_modes(self._h, out), "tm_engine_get_job_modes")
        return np.array(out, np.int32).reshape(6, 3)

    def uses_fused_edge(self, n_slots: Optional[int] = None) -> bool:
        """does a compute of n_slots slots (default: the batch capacity) run its edge-only jobs in the fused kernel?"""
        r = self._L.tm_engine_uses_fused_edge(self._h, int(self.batch if n_slots is None else n_slots))
        _chk(min(r, 0), "tm_engine_uses_fused_edge")
        return r > 0

    def debug_set_edge_beside(self, mode: int):
        """measurement hook: 1 = the fused kernel of the edge-only jobs beside the two blur passes (default), 0 = behind them"""
        _chk(self._L.tm_engine_debug_set_edge_beside(self._h, int(mode)), "tm_engine_debug_set_edge_beside")

    def debug_set_param(self, param: int, value: int):
        """tuning values / fault injection of this engine (ffi.TM_DBG_*; see the header)"""
        _chk(self._L.tm_engine_debug_set_param(self._h, int(param), int(value)), "tm_engine_debug_set_param")

    def debug_chain(self, peer):
        """measurement hook: this engine's ingest waits for `peer`'s column pass, its column pass for `peer`'s row pass (None unchains)"""
        _chk(self._L.tm_engine_debug_chain(self._h, peer._h if peer is not None else None), "tm_engine_debug_chain")

    def debug_set_edge_epoch(self, epoch: int):
        _chk(self._L.tm_engine_debug_set_edge_epoch(self._h, int(epoch)), "tm_engine_debug_set_edge_epoch")

    def set_graph(self, on: Optional[bool]):
        """True / False: always / never replay the launch sequence from a captured hipGraph; None: the default (small launches that
        repeat replay by themselves)"""
        self._retire_deferred()
        _chk(self._L.tm_engine_set_graph(self._h, -1 if on is None else int(bool(on))), "tm_engine_set_graph")
        self._peer_follows("set_graph", on)

    def set_variant(self, v: int):
        self._retire_deferred()
        _chk(self._L.tm_engine_set_variant(self._h, int(v)), "tm_engine_set_variant")
        self._peer_follows("set_variant", v)

    def read_plane(self, slot: int, kind: int, scale: int, index: int, channel: int) -> np.ndarray:
        w, h = self.width, self.height
        for _ in range(scale):
            w, h = (w + 1) // 2, (h + 1) // 2
        transposed = kind in (ffi.TM_PLANE_XYB_T, ffi.TM_PLANE_PASS1_T)
        out = np.zeros((w, h) if transposed else (h, w), np.float32)
        _chk(self._L.tm_engine_debug_read_plane(self._h, slot, kind, scale, index, channel,
                                                out.ctypes.data_as(C.POINTER(C.c_float)), out.size), "tm_engine_debug_read_plane")
        return out


class Ssimulacra2:
    """Mirror of ssimulacra2_cuda::Ssimulacra2: inputs are LINEAR RGB f32 images (h, w, 3)."""

    def __init__(self, width: int, height: int):
        self._tm = TurboMetrics(width, height, Metrics(ssimulacra2=True), batch=1)

    def compute_sync(self, ref_linear, dis_linear) -> float:
        return self._tm.compute_one(HwFrame.linear(ref_linear), HwFrame.linear(dis_linear)).ssimulacra2

    def compute_srgb_sync(self, ref_srgb8, dis_srgb8) -> float:
        """== compute_from_cpu_srgb_sync (lib.rs:232-266): packed sRGB u8 through the LUT."""
        return self._tm.compute_one(HwFrame.rgb(np.asarray(ref_srgb8, np.uint8)), HwFrame.rgb(np.asarray(dis_srgb8, np.uint8))).ssimulacra2

    def mem_usage(self) -> int:
        return self._tm.mem_usage()

    def close(self):
        self._tm.close()


def score_from_sums(sums, width, height) -> float:
    s = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    return float(ffi.lib().tm_ssimulacra2_score_from_sums(s.ctypes.data_as(C.POINTER(C.c_double)), int(width), int(height)))


def ssim_from_sums(sums, width, height) -> float:
    s = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    return float(ffi.lib().tm_ssim_from_sums(s.ctypes.data_as(C.POINTER(C.c_double)), int(width), int(height)))


def msssim_from_sums(sums, width, height) -> float:
    s = np.ascontiguousarray(np.asarray(sums, np.float64).ravel())
    return float(ffi.lib().tm_msssim_from_sums(s.ctypes.data_as(C.POINTER(C.c_double)), int(width), int(height)))
