"""Seeded synthetic frame-pair generators for the benchmark and the parity tests.

Nothing here is on the measured path: frames are generated on the host with numpy, then handed to
the engine through the C ABI exactly like decoded frames would be.

Surfaces follow the reference's decoded-frame contract (cudarse-video/src/dec.rs:299-393): one
allocation, luma plane of `coded_height` rows at `pitch` bytes, then the interleaved CbCr plane at
`pitch * coded_height`; NV12 = u8, P016 = u16 with the 10-bit value in the HIGH bits.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Counter-based hash (vectorised splitmix64 finaliser) -> uint64."""
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _noise(seed, shape, plane, amp):
    """uniform integers in [-amp, amp], a pure function of (seed, plane, y, x)."""
    h, w = shape
    yy, xx = np.meshgrid(np.arange(h, dtype=np.uint64), np.arange(w, dtype=np.uint64), indexing="ij")
    key = np.uint64(seed) ^ ((np.uint64(plane) << np.uint64(40)) + (yy << np.uint64(20)) + xx)
    return (splitmix64(key) % np.uint64(2 * amp + 1)).astype(np.int64) - amp


def yuv420_pair(w, h, n, bits=8):
    """Planar 4:2:0 reference/distorted pair number `n` (limited range), values in `bits` precision.
    Returns ((Y,Cb,Cr), (Y,Cb,Cr)) integer arrays; chroma planes are ceil(h/2) x ceil(w/2)."""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    cy, cx = np.meshgrid(np.arange(ch) * 2, np.arange(cw) * 2, indexing="ij")
    two_pi = 2.0 * np.pi
    luma = 16 + 219 * (0.5 + 0.25 * np.sin(two_pi * (3.0 * x / w + n / 32.0)) + 0.2 * np.cos(two_pi * 2.0 * y / h))
    cb = 128 + 60 * np.sin(two_pi * (cx / w + cy / h) + n / 16.0)
    cr = 128 + 60 * np.cos(two_pi * (cx / w + cy / h) + n / 16.0)
    seed = 0xC0FFEE00 + n
    Y = np.clip(np.rint(luma) + _noise(seed, (h, w), 0, 8), 16, 235).astype(np.int64)
    Cb = np.clip(np.rint(cb) + _noise(seed, (ch, cw), 1, 4), 16, 240).astype(np.int64)
    Cr = np.clip(np.rint(cr) + _noise(seed, (ch, cw), 2, 4), 16, 240).astype(np.int64)
    # distorted: luma quantised to multiples of 6, 5-tap vertical box blur on chroma, a little noise
    dseed = 0xC0FFEE80 + n
    Yd = np.clip((Y // 6) * 6 + _noise(dseed, (h, w), 0, 1), 16, 235)

    def vblur(p):
        pad = np.pad(p, ((2, 2), (0, 0)), mode="edge")
        return (pad[0:-4] + pad[1:-3] + pad[2:-2] + pad[3:-1] + pad[4:]) // 5

    Cbd = np.clip(vblur(Cb) + _noise(dseed, (ch, cw), 1, 1), 16, 240)
    Crd = np.clip(vblur(Cr) + _noise(dseed, (ch, cw), 2, 1), 16, 240)
    if bits == 10:
        up = lambda p: p * 4 + (_noise(seed ^ 0x55, p.shape, 3, 1) + 1)  # exercise the 2 extra bits
        ref = tuple(up(p) for p in (Y, Cb, Cr))
        dis = tuple(up(p) for p in (Yd, Cbd, Crd))
        return ref, dis
    return (Y, Cb, Cr), (Yd, Cbd, Crd)


def pack_biplanar(planes, w, h, bits=8, pitch=None, coded_height=None):
    """(Y,Cb,Cr) -> one NV12 (bits=8) or P016 (bits=10, MSB aligned in u16) surface.
    Returns (surface uint8 1-D, pitch bytes, coded_height)."""
    Y, Cb, Cr = planes
    bps = 1 if bits == 8 else 2
    cw, ch = (w + 1) // 2, (h + 1) // 2
    if pitch is None:
        pitch = ((max(w, 2 * cw) * bps + 255) // 256) * 256
    if coded_height is None:
        coded_height = ((h + 15) // 16) * 16
    dt = np.uint8 if bits == 8 else np.uint16
    shift = 0 if bits == 8 else 16 - bits
    surf = np.zeros((coded_height + (coded_height + 1) // 2, pitch // bps), dt)
    surf[:h, :w] = (Y.astype(np.int64) << shift).astype(dt)
    uv = np.zeros((ch, 2 * cw), dt)
    uv[:, 0::2] = (Cb.astype(np.int64) << shift).astype(dt)
    uv[:, 1::2] = (Cr.astype(np.int64) << shift).astype(dt)
    surf[coded_height:coded_height + ch, :2 * cw] = uv
    return surf.view(np.uint8).reshape(-1), pitch, coded_height


P10_RUN = 128  # tm_geom.h TM_P10_RUN: samples per contiguous run = words per block of a TM_KIND_I420_P10 row


def p10_row_words(n):
    return -(-n // (3 * P10_RUN)) * P10_RUN


def p10_pack_plane(plane, pitch_words=None):
    """(rows, n) integers in [0, 1023] -> (rows, pitch_words) uint32 in the packed 10-bit upload layout (include/turbo_metrics_hip.h,
    tm_engine_set_frame_i420p10): word k of block b = s[384 b + k] | s[384 b + 128 + k] << 10 | s[384 b + 256 + k] << 20, absent samples 0.
    Written with numpy from the header's sentence, not through the library's tm_p10_pack_rows (the tests hold the two against each other)."""
    plane = np.asarray(plane)
    rows, n = plane.shape
    words = p10_row_words(n)
    pitch_words = words if pitch_words is None else pitch_words
    pad = np.zeros((rows, words // P10_RUN, 3, P10_RUN), np.uint32)
    pad.reshape(rows, -1)[:, :n] = plane.astype(np.uint32) & 1023
    out = np.zeros((rows, pitch_words), np.uint32)
    out[:, :words] = (pad[:, :, 0] | (pad[:, :, 1] << 10) | (pad[:, :, 2] << 20)).reshape(rows, words)
    return out


def p10_unpack_plane(words, n):
    """inverse of p10_pack_plane: (rows, >= p10_row_words(n)) uint32 -> (rows, n) int64"""
    words = np.asarray(words, np.uint32)
    rows = words.shape[0]
    w = words[:, :p10_row_words(n)].reshape(rows, -1, 1, P10_RUN)
    s = np.concatenate([(w >> np.uint32(10 * j)) & np.uint32(1023) for j in range(3)], axis=2)
    return s.reshape(rows, -1)[:, :n].astype(np.int64)


def nv12_pair(w, h, n):
    ref, dis = yuv420_pair(w, h, n, 8)
    return pack_biplanar(ref, w, h, 8), pack_biplanar(dis, w, h, 8)


def p016_pair(w, h, n):
    ref, dis = yuv420_pair(w, h, n, 10)
    return pack_biplanar(ref, w, h, 10), pack_biplanar(dis, w, h, 10)


def rgb8_pair(w, h, seed=0x5EED0001):
    """Config-1 style packed RGB8 pair: smooth pattern + hashed noise; distorted = 3-tap horizontal
    box blur, quantised to multiples of 4, plus +-2 noise.  Returns two (h, w, 3) uint8 arrays."""
    y, x = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    ref = np.zeros((h, w, 3), np.int64)
    for c in range(3):
        base = 128 + 80 * np.sin(2 * np.pi * 3 * x / w + c) * np.cos(2 * np.pi * 2 * y / h)
        ref[..., c] = np.clip(np.rint(base) + _noise(seed, (h, w), c, 12), 0, 255)
    pad = np.pad(ref, ((0, 0), (1, 1), (0, 0)), mode="edge")
    blur = (pad[:, :-2] + pad[:, 1:-1] + pad[:, 2:]) // 3
    dis = np.zeros_like(ref)
    for c in range(3):
        dis[..., c] = np.clip((blur[..., c] // 4) * 4 + _noise(seed + 1, (h, w), c, 2), 0, 255)
    return ref.astype(np.uint8), dis.astype(np.uint8)
