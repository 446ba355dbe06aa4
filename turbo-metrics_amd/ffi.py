"""ctypes binding of the C ABI (include/turbo_metrics_hip.h + the laboratory half, include/turbo_metrics_hip_debug.h).  Loads the in-tree
LABORATORY build, lab/libturbometrics_hip_lab.so -- the engine of libturbometrics_hip.so (the ship build: facade only, what the CLI links)
plus kernel variants, stage timers, plane read-back and fault injection, which tests/, tools/ and bench.py's roofline measurement need; the
kernels the two builds share are the same device code (tests/test_abi_symbols.py).  Raises loudly if it is missing -- there is no fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TM_HIP_LIB") or os.path.join(_HERE, "lab", "libturbometrics_hip_lab.so")  # override: experiment builds only
SHIP_LIB_PATH = os.path.join(_HERE, "libturbometrics_hip.so")

TM_OK, TM_ERR_INVALID_ARG, TM_ERR_UNSUPPORTED, TM_ERR_HIP, TM_ERR_OOM, TM_ERR_STATE = range(6)
TM_METRIC_PSNR, TM_METRIC_SSIM, TM_METRIC_MSSSIM, TM_METRIC_SSIMULACRA2 = 1, 2, 4, 8
TM_MATRIX_BT709, TM_MATRIX_BT601_525, TM_MATRIX_BT601_625 = 0, 1, 2
TM_TRANSFER_BT709 = 0
TM_SIDE_REF, TM_SIDE_DIS = 0, 1
TM_CHANNELS_POOLED, TM_CHANNELS_FIRST = 0, 1
TM_MEM_HOST, TM_MEM_DEVICE, TM_MEM_HOST_PINNED = 0, 1, 2
TM_STAGE_INGEST, TM_STAGE_BLUR_V, TM_STAGE_BLUR_H, TM_STAGE_SSIM, TM_STAGE_EDGE, TM_STAGE_COUNT = 0, 1, 2, 3, 4, 5
TM_PLANE_LINEAR, TM_PLANE_XYB, TM_PLANE_XYB_T, TM_PLANE_PASS1_T = 0, 1, 2, 3
TM_VARIANT_DEFAULT, TM_VARIANT_REFERENCE, TM_VARIANT_WIDE_ROWS, TM_VARIANT_TILE_INGEST, TM_VARIANT_SPLIT_ROWS, TM_VARIANT_WHOLE_ROWS = 0, 1, 0x100, 0x200, 0x400, 0x800
TM_VARIANT_TWO_PASS_EDGE, TM_VARIANT_UPPER_KERNEL, TM_VARIANT_FUSED_EDGE = 0x1000, 0x2000, 0x4000
TM_DBG_FUSED_EDGE_FROM, TM_DBG_EF_WAVES, TM_DBG_EF_PERSIST_WGS, TM_DBG_PASS_PRIO, TM_DBG_SPLIT_ROWS_BELOW, TM_DBG_SOLO_COL_BELOW, TM_DBG_EF_FAULT, TM_DBG_LINEAR_UPLOAD, TM_DBG_UPLOAD_STREAMS, TM_DBG_UPLOAD_MERGE = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9


class FrameScoresC(C.Structure):
    _fields_ = [("psnr", C.c_double), ("ssim", C.c_double), ("msssim", C.c_double),
                ("ssimulacra2", C.c_double), ("valid", C.c_uint32)]


# every symbol the two headers declare: name -> (restype, argtypes)
_vp, _u32, _i, _sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
SYMBOLS = {
    "tm_init": (_i, [_i]),
    "tm_device_count": (_i, []),
    "tm_device_mem_info": (_i, [C.POINTER(_sz), C.POINTER(_sz)]),
    "tm_device_numa_node": (_i, [_i]),
    "tm_host_alloc": (_vp, [_sz]),
    "tm_host_free": (None, [_vp]),
    "tm_set_placement_candidates": (None, [_i]),
    "tm_set_debug_log": (None, [_i]),
    "tm_engine_create": (_i, [C.POINTER(_vp), _u32, _u32, _u32, _u32]),
    "tm_engine_destroy": (None, [_vp]),
    "tm_engine_mem_usage": (_sz, [_vp]),
    "tm_engine_set_frame_nv12": (_i, [_vp, _u32, _i, _vp, _vp, _sz, _i, _i, _i, _i]),
    "tm_engine_set_frame_p016": (_i, [_vp, _u32, _i, _vp, _vp, _sz, _i, _i, _i, _i]),
    "tm_engine_set_surface_nv12": (_i, [_vp, _u32, _i, _vp, _sz, _u32, _i, _i, _i, _i]),
    "tm_engine_set_surface_p016": (_i, [_vp, _u32, _i, _vp, _sz, _u32, _i, _i, _i, _i]),
    "tm_engine_set_frame_i420": (_i, [_vp, _u32, _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _i, _i, _i]),
    "tm_engine_set_frame_i420p10": (_i, [_vp, _u32, _i, _vp, _vp, _vp, _sz, _sz, _i, _i, _i, _i]),
    "tm_p10_row_bytes": (_sz, [_u32]),
    "tm_p10_pack_rows": (None, [_vp, _sz, _u32, _u32, _vp, _sz]),
    "tm_engine_set_frame_rgb8": (_i, [_vp, _u32, _i, _vp, _sz, _i]),
    "tm_engine_set_frame_rgb16": (_i, [_vp, _u32, _i, _vp, _sz, _i]),
    "tm_engine_set_frame_rgbf32": (_i, [_vp, _u32, _i, _vp, _sz, _i]),
    "tm_engine_set_frame_linear_f32": (_i, [_vp, _u32, _i, _vp, _sz, _i]),
    "tm_engine_upload_fence": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "tm_engine_upload_done": (_i, [_vp, C.c_uint64, _i]),
    "tm_engine_compute_async": (_i, [_vp, _u32]),
    "tm_engine_sync": (_i, [_vp]),
    "tm_engine_get_scores": (_i, [_vp, _u32, C.POINTER(FrameScoresC)]),
    "tm_engine_get_scores_batch": (_i, [_vp, _u32, _u32, C.POINTER(FrameScoresC)]),
    "tm_engine_get_raw_sums": (_i, [_vp, _u32, C.POINTER(C.c_double)]),
    "tm_engine_set_full_sums": (_i, [_vp, _i]),
    "tm_engine_get_job_modes": (_i, [_vp, C.POINTER(C.c_int)]),
    "tm_engine_uses_fused_edge": (_i, [_vp, C.c_uint32]),
    "tm_engine_get_sse": (_i, [_vp, _u32, C.POINTER(C.c_uint64)]),
    "tm_engine_get_sse_channels": (_i, [_vp, _u32, C.POINTER(C.c_uint64)]),
    "tm_psnr_from_sse": (C.c_double, [C.c_uint64, C.c_uint64]),
    "tm_engine_set_channel_mode": (_i, [_vp, _i]),
    "tm_ssim_channel_from_sums": (C.c_double, [C.POINTER(C.c_double), _u32, _u32, _i]),
    "tm_msssim_channel_from_sums": (C.c_double, [C.POINTER(C.c_double), _u32, _u32, _i]),
    "tm_engine_get_ssim_sums": (_i, [_vp, _u32, C.POINTER(C.c_double)]),
    "tm_ssim_from_sums": (C.c_double, [C.POINTER(C.c_double), _u32, _u32]),
    "tm_msssim_from_sums": (C.c_double, [C.POINTER(C.c_double), _u32, _u32]),
    "tm_ssim_window": (None, [C.POINTER(C.c_float)]),
    "tm_ssimulacra2_score_from_sums": (C.c_double, [C.POINTER(C.c_double), _u32, _u32]),
    "tm_engine_set_profiling": (_i, [_vp, _i]),
    "tm_engine_get_stage_ms": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), _i]),
    "tm_engine_set_graph": (_i, [_vp, _i]),
    "tm_engine_set_linear_upload": (_i, [_vp, _i]),
    "tm_engine_set_variant": (_i, [_vp, _i]),
    "tm_engine_debug_set_v_offset": (_i, [_vp, _sz]),
    "tm_engine_debug_set_ingest_rows": (_i, [_vp, _i]),
    "tm_engine_debug_set_edge_beside": (_i, [_vp, _i]),
    "tm_engine_debug_set_edge_epoch": (_i, [_vp, C.c_uint32]),
    "tm_engine_debug_set_param": (_i, [_vp, _i, C.c_longlong]),
    "tm_engine_debug_chain": (_i, [_vp, _vp]),
    "tm_engine_debug_read_plane": (_i, [_vp, _u32, _i, _i, _i, _i, C.POINTER(C.c_float), _sz]),
    "tm_strerror": (C.c_char_p, [_i]),
    "tm_last_hip_error": (C.c_char_p, []),
    "tm_version": (C.c_char_p, []),
}

_lib = None


def lib():
    """The loaded shared library with typed entry points."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
