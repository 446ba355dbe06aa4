#!/usr/bin/env python3
"""Mutation fuzz of the host-side frame sources (PNG incl. 16-bit / Adam7, PPM, PFM, Y4M 8/10-bit, Matroska / IVF demuxing) through the sanitised test
helper: TM_HOST_SANITIZE=1 python -m pytest tests/test_host_cli.py -m "not gpu" builds tests/host/tm_host_test_san
(AddressSanitizer + UBSan, abort on report).  Every mutated file must end in exit code 0 (decoded) or 1 (clean error):
anything else -- a sanitizer report, a signal -- is printed and counted.  usage: tools/fuzz_sources.py [iterations per seed file]"""
import os, random, struct, subprocess, sys, tempfile, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "host", "tm_host_test_san")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(1234)
nrng = np.random.default_rng(7)


def png(w, h, depth, interlace):
    ch = 3
    raw = nrng.integers(0, 256, (h, w * ch * depth // 8), dtype=np.uint8)
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    if interlace:  # Adam7: the seven passes, each as its own scanlines
        bpp = ch * depth // 8
        img = raw.reshape(h, w, bpp)
        data = b""
        for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                data += b"".join(b"\0" + sub[r].tobytes() for r in range(sub.shape[0]))
    else:
        data = b"".join(b"\0" + raw[r].tobytes() for r in range(h))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 2, 0, 0, interlace)) + chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b"")


def apng(w, h, depth, n):
    """animated PNG: a full first frame (IDAT) and n - 1 sub-rectangle frames (fdAT) with the three dispose operations"""
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    bpp = 3 * depth // 8
    def z(ww, hh): return zlib.compress(b"".join(b"\0" + nrng.integers(0, 256, ww * bpp, dtype=np.uint8).tobytes() for _ in range(hh)))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 2, 0, 0, 0)) + chunk(b"acTL", struct.pack(">II", n, 0))
    out += chunk(b"fcTL", struct.pack(">IIIIIHHBB", 0, w, h, 0, 0, 1, 10, 0, 0)) + chunk(b"IDAT", z(w, h))
    seq = 1
    for k in range(1, n):
        fw, fh, x, y = max(1, w // 2), max(1, h // 3), k % max(1, w // 2), k % max(1, h // 2)
        out += chunk(b"fcTL", struct.pack(">IIIIIHHBB", seq, fw, fh, x, y, 1, 10, k % 3, k % 2)) + chunk(b"fdAT", struct.pack(">I", seq + 1) + z(fw, fh))
        seq += 2
    return out + chunk(b"IEND", b"")


seeds = {
    "a.png": png(37, 23, 8, 0), "b.png": png(19, 11, 16, 0), "c.png": png(21, 13, 8, 1), "m.png": apng(29, 17, 8, 5), "n.png": apng(13, 9, 16, 3),
    "d.ppm": b"P6\n17 9\n255\n" + nrng.integers(0, 256, 17 * 9 * 3, dtype=np.uint8).tobytes(),
    "e.ppm": b"P6\n# c\n9 5 65535\n" + nrng.integers(0, 256, 9 * 5 * 6, dtype=np.uint8).tobytes(),
    "f.pfm": b"PF\n8 6\n-1.0\n" + nrng.random(8 * 6 * 3, dtype=np.float32).tobytes(),
    "g.y4m": b"YUV4MPEG2 W70 H38 F30:1 Ip A1:1 C420jpeg\n" + b"".join(b"FRAME\n" + nrng.integers(0, 256, 70 * 38 + 2 * 35 * 19, dtype=np.uint8).tobytes() for _ in range(2)),
    "h.y4m": b"YUV4MPEG2 W46 H30 C420p10 XCOLORRANGE=FULL\n" + b"".join(b"FRAME\n" + nrng.integers(0, 1024, 46 * 30 + 2 * 23 * 15, dtype=np.uint16).tobytes() for _ in range(2)),
}
# compressed-video containers (video_input.cpp): demuxed and their sequence headers parsed; the files are built by the test writers
sys.path.insert(0, ROOT)
from tests import test_video_input as V  # noqa: E402
_sps, _pps = V.h264_sps(640, 360, vui=(0, 1, 1, 1)), bytes([0x68, 0xCE, 0x3C, 0x80])
_nal = [bytes([0x65, 1, 2, 3, i]) + bytes(range(20)) for i in range(6)]
_seq = V.av1_sequence_header_obu(320, 240, color=(1, 1, 1))
seeds["i.mkv"] = V.mkv_file(b"V_MPEG4/ISO/AVC", V.avcc(_sps, _pps), [(2, struct.pack(">I", len(n)) + n) for n in _nal], 640, 360)
seeds["j.mkv"] = V.mkv_file(b"V_AV1", bytes([0x81, 8, 12, 0]) + _seq, [(2, bytes([6 << 3 | 2, 2, i, i])) for i in range(6)], 320, 240, unknown_cluster_size=True)
seeds["k.ivf"] = V.ivf_file(b"AV01", 320, 240, [bytes([(2 << 3) | 2, 0]) + _seq + bytes([6 << 3 | 2, 1, 7])] + [bytes([6 << 3 | 2, 2, i, i]) for i in range(4)])
seeds["l.mkv"] = V.mkv_file(b"V_MPEG2", V.mpeg2_sequence(720, 480, color=(6, 6, 6)), [(2, [b"\0\0\1\0ab", b"\0\0\1\0cd"]), (2, b"\0\0\1\0zz")], 720, 480, lacing="fixed")
bad = 0
runs = 0
with tempfile.TemporaryDirectory() as d:
    for name, blob in seeds.items():
        for it in range(N + 1):
            b = bytearray(blob)
            if it:  # iteration 0 = the unmodified seed (must decode)
                kind = rng.randrange(4)
                if kind == 0:
                    for _ in range(rng.randrange(1, 6)): b[rng.randrange(len(b))] = rng.randrange(256)
                elif kind == 1: b = b[: rng.randrange(1, len(b))]
                elif kind == 2:
                    i = rng.randrange(len(b)); b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
                else:  # mutate inside the first 64 bytes (headers)
                    for _ in range(rng.randrange(1, 4)): b[rng.randrange(min(64, len(b)))] = rng.choice(b"0123456789 \n\xff\x00W")
            p = os.path.join(d, name)
            open(p, "wb").write(bytes(b))
            runs += 1
            try:
                # video containers: the demuxer + sequence-header parsers (`demux` writes the whole elementary stream); everything else: `source`
                r = subprocess.run([HELPER, "demux" if name.endswith((".mkv", ".ivf")) else "source", p, os.path.join(d, "out.bin")], capture_output=True, text=True, errors="replace", timeout=20,
                                   env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
            except subprocess.TimeoutExpired:
                bad += 1
                keep = os.path.join(tempfile.gettempdir(), f"fuzz_hang_{it}_{name}")
                open(keep, "wb").write(bytes(b))
                print("HANG", name, "iteration", it, "-> kept as", keep)
                continue
            if r.returncode not in (0, 1) or (it == 0 and r.returncode != 0):
                bad += 1
                print("FAIL", name, "iteration", it, "rc", r.returncode, "|", r.stdout[-120:].strip(), "|", r.stderr[-400:].strip())
print(f"{runs} runs, {bad} failures")
sys.exit(1 if bad else 0)
