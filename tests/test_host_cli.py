"""C++ host side (turbo-metrics_amd/host): number formatting, Stats, frame sources and the output layer on the CPU tier
(through tests/host/tm_host_test); the `turbo-metrics` command line itself on the GPU tier.

Expected texts follow the reference's output layer (crates/turbo-metrics-cli/src/output.rs, quick-stats/src/lib.rs):
Rust `{}` for CSV, serde_json (ryu) for JSON, `{:#?}` for the default report."""
import json
import os
import struct
import subprocess
import sys
import zlib
from decimal import Decimal

import numpy as np
import pytest

from tm_pkg import tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "turbo-metrics_amd", "host")
# TM_HOST_SANITIZE=1: the helper (decoders, stream readers, formatters) built with AddressSanitizer + UBSan, aborting on the first report;
# TM_HOST_SANITIZE=thread: with ThreadSanitizer (the read-ahead pool, the row workers, the ring helper)
SANITIZE = os.environ.get("TM_HOST_SANITIZE") in ("1", "thread")
TSAN = os.environ.get("TM_HOST_SANITIZE") == "thread"
HELPER = os.path.join(ROOT, "tests", "host", "tm_host_test_tsan" if TSAN else "tm_host_test_san" if SANITIZE else "tm_host_test")
CLI = os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")


@pytest.fixture(scope="module")
def helper():
    srcs = [os.path.join(ROOT, "tests", "host", "tm_host_test.cpp")] + [os.path.join(HOST, f) for f in ("frame_sources.cpp", "video_input.cpp", "output.cpp", "turbo_metrics.cpp", "ranks.cpp")]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")]
    if not os.path.exists(HELPER) or any(os.path.getmtime(d) > os.path.getmtime(HELPER) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17"] + (["-fsanitize=thread", "-fno-omit-frame-pointer", "-g"] if TSAN else ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"] if SANITIZE else [])
                              + ["-o", HELPER] + srcs + ["-L" + os.path.join(ROOT, "turbo-metrics_amd"),
                              "-lturbometrics_hip", "-lz", "-ldl", "-pthread", "-Wl,-rpath," + os.path.join(ROOT, "turbo-metrics_amd")])
    return HELPER


PACK10 = "0"  # the source tests below look at 10-bit pictures as the stream's own 16-bit words; the packing tests set "1" (the product's default)


def run(helper, *args, stdin=None):
    r = subprocess.run([helper] + [str(a) for a in args], input=stdin, capture_output=True, text=True, check=False, env=dict(os.environ, TM_PACK10=PACK10))
    if SANITIZE:  # a report of a sanitizer fails the test that provoked it, whatever the helper printed
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.stdout


# ---- number formatting ---------------------------------------------------------------------------------------------
def rust_display(x):
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    s = format(Decimal(repr(x)), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s if x != 0 else ("-0" if str(x).startswith("-") else "0")


def test_f64_text_forms_follow_rust_display_debug_and_serde_json(helper):
    rng = np.random.default_rng(3)
    vals = [100.0, 0.1, 74.54997609000992, 17.398505, 1e-7, 1e16, 9999999999999998.0, 1e15, 123456.789, 0.0001, 0.00009999, 5e-324,
            35.73, 99.99, 1.0, 2.5, 1e21, 0.00001, 0.000001, 1e-5 * 0.99, 12345678.9] + list(rng.random(40) * 100) + list(10.0 ** rng.uniform(-9, 19, 40))
    out = run(helper, "fmt", *[repr(float(v)) for v in vals]).strip().split("\n")
    assert len(out) == len(vals)
    for v, line in zip(vals, out):
        v = float(v)
        disp, dbg, js = line.split("|")
        assert disp == rust_display(v), (v, disp)
        assert float(disp) == v and "e" not in disp  # Display never uses an exponent and round-trips
        assert float(dbg) == v and float(js) == v    # shortest round-trip digits everywhere
        assert json.loads(js) == v
        a = abs(v)
        assert ("e" in dbg) == (a < 1e-4 or a >= 1e16), (v, dbg)        # core::fmt float_to_general_debug
        if "e" not in dbg:
            assert "." in dbg                                            # Debug keeps ".0"
        assert ("e" in js) == (not (1e-5 <= a < 1e16)), (v, js)  # ryu: decimal iff -5 < kk <= 16
        if "e" not in js:
            assert "." in js
    assert run(helper, "fmt", "inf", "nan", "0", "-0.0").strip().split("\n") == ["inf|inf|null", "NaN|NaN|null", "0|0.0|0.0", "-0|-0.0|-0.0"]
    # exact forms of a few known values
    assert run(helper, "fmt", "100", "1e-7", "1e16", "1.5e-5").strip().split("\n") == [
        "100|100.0|100.0", "0.0000001|1e-7|1e-7", "10000000000000000|1e16|1e16", "0.000015|1.5e-5|0.000015"]


def test_stats_match_quick_stats_definitions(helper):
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 100, 1001):
        v = rng.random(n) * 100
        got = json.loads(run(helper, "stats", *[repr(float(x)) for x in v]))
        s = np.sort(v)
        assert got["min"] == s[0] and got["max"] == s[-1]
        np.testing.assert_allclose(got["mean"], v.mean(), rtol=1e-14)
        np.testing.assert_allclose(got["var"], v.var() if n > 1 else 0.0, rtol=1e-12, atol=0)
        np.testing.assert_allclose(got["sample_var"], v.var(ddof=1) if n > 1 else 0.0, rtol=1e-12, atol=0)
        np.testing.assert_allclose(got["stddev"], np.sqrt(got["var"]), rtol=1e-15)
        for k, p in (("p1", 1), ("p5", 5), ("p50", 50), ("p95", 95), ("p99", 99)):
            np.testing.assert_allclose(got[k], np.percentile(v, p), rtol=1e-13)  # linear interpolation on (n-1)*p/100
    # the reference's own example values (quick-stats/src/lib.rs:101-107)
    assert json.loads(run(helper, "stats", 0, 1, 3, 4))["mean"] == 2.0
    assert run(helper, "stats").startswith("ERROR")  # empty input: the reference panics (index out of bounds)


# ---- output layer --------------------------------------------------------------------------------------------------
SCORES = "35.5 80.25\n100 99.99\n"


def test_output_csv_repeats_header_and_rows_like_the_reference(helper):
    out = run(helper, "output", "csv", "1001", stdin=SCORES)
    # prepare(): header; per frame: row; output_results(): header + every row again (output.rs:25-37,104-139)
    assert out == "psnr,ssimulacra2\n35.5,80.25\n100,99.99\npsnr,ssimulacra2\n35.5,80.25\n100,99.99\n"


def test_output_json_lines(helper):
    out = run(helper, "output", "json-lines", "1001", stdin=SCORES).strip().split("\n")
    assert out[0] == '{"psnr":35.5,"ssimulacra2":80.25}' and out[1] == '{"psnr":100.0,"ssimulacra2":99.99}'
    last = json.loads(out[2])
    assert list(last) == ["frame_count", "psnr", "ssimulacra2"] and last["frame_count"] == 2
    assert list(last["psnr"]) == ["min", "max", "mean", "var", "sample_var", "stddev", "sample_stddev", "p1", "p5", "p50", "p95", "p99"]
    assert out[2].startswith('{"frame_count":2,"psnr":{"min":35.5,"max":100.0,"mean":67.75,')


def test_output_json_pretty(helper):
    out = run(helper, "output", "json", "0001", stdin="80.25\n99.99\n")
    assert out.startswith('{\n  "frame_count": 2,\n  "ssimulacra2": {\n    "scores": [\n      80.25,\n      99.99\n    ],\n    "stats": {\n      "min": 80.25,\n')
    assert out.endswith('      "p99": ' + json.dumps(json.loads(out)["ssimulacra2"]["stats"]["p99"]) + "\n    }\n  }\n}\n")
    d = json.loads(out)
    assert d["frame_count"] == 2 and d["ssimulacra2"]["scores"] == [80.25, 99.99]


def test_output_default_is_rust_pretty_debug(helper):
    out = run(helper, "output", "default", "0001", stdin="80.25\n99.75\n")
    assert out == ("SSIMULACRA2: Stats {\n    min: 80.25,\n    max: 99.75,\n    mean: 90.0,\n    var: 95.0625,\n    sample_var: 190.125,\n"
                   "    stddev: 9.75,\n    sample_stddev: 13.788582233137676,\n    p1: 80.445,\n    p5: 81.225,\n    p50: 90.0,\n    p95: 98.775,\n    p99: 99.555,\n}\n")


def test_psnr_of_identical_frames_prints_inf_and_null(helper):
    assert run(helper, "output", "csv", "1000", stdin="inf\n").split("\n")[1] == "inf"
    assert run(helper, "output", "json-lines", "1000", stdin="inf\n").split("\n")[0] == '{"psnr":null}'


# ---- colour metadata -------------------------------------------------------------------------------------------------
def test_colour_fallback_and_todo_combinations(helper):
    assert run(helper, "colors", 2, 2, 2, 480).split() == ["BT601_525", "BT601_525", "BT709", "1", "0"]   # color.rs:51-78
    assert run(helper, "colors", 2, 2, 2, 576).split() == ["BT601_625", "BT601_625", "BT709", "2", "0"]
    assert run(helper, "colors", 2, 2, 2, 1080).split() == ["BT709", "BT709", "BT709", "0", "0"]
    assert run(helper, "colors", 2, 2, 2, 2160).split() == ["BT709", "BT709", "BT709", "0", "0"]
    assert run(helper, "colors", 1, 1, 1, 480).split() == ["BT709", "BT709", "BT709", "0", "0"]
    assert run(helper, "colors", 5, 5, 6, 1080).split()[:3] == ["BT601_625", "BT601_625", "BT709"]
    assert "not implemented" in run(helper, "colors", 1, 6, 1, 1080)   # mixed primaries / matrix: todo!() at color.rs:85
    assert "not implemented" in run(helper, "colors", 1, 1, 4, 1080)   # gamma 2.2 transfer: todo!() at color.rs:92


# ---- frame sources ---------------------------------------------------------------------------------------------------
def png_bytes(arr, interlace=False):
    """Minimal PNG writer (RGB 8/16 bit, filter 0, optional Adam7) -- Pillow cannot write 16-bit RGB or interlaced files."""
    h, w, _ = arr.shape
    depth = 8 if arr.dtype == np.uint8 else 16
    be = arr if depth == 8 else arr.astype(">u2")

    def rows(sub):
        return b"".join(b"\x00" + sub[y].tobytes() for y in range(sub.shape[0]))
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = be[y0::dy, x0::dx]
            if sub.size:
                raw += rows(sub)
    else:
        raw = rows(be)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    ihdr = struct.pack(">IIBBBBB", w, h, depth, 2, 0, 0, 1 if interlace else 0)
    comp = zlib.compress(raw, 6)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")


def read_dump(helper, path, out, *extra):
    txt = run(helper, "source", path, out, *extra).strip().split("\n")
    return txt[0].split(), [t.split() for t in txt[1:]], (np.fromfile(out, np.uint8) if os.path.exists(out) else None)


def test_png_sources_decode_to_packed_rgb(helper, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    a8 = (rng.random((37, 53, 3)) * 255).astype(np.uint8)
    a8[5:20, 3:30] = (np.arange(27)[None, :, None] * 9 + np.arange(15)[:, None, None] * 3).astype(np.uint8)  # smooth area: PNG filters 1-4
    p = str(tmp_path / "a.png")
    Image.fromarray(a8).save(p, optimize=True)
    head, frames, data = read_dump(helper, p, str(tmp_path / "a.bin"))
    assert head[:3] == ["PNG/turbo-metrics-hip", "53", "37"] and head[3:7] == ["BT709", "BT709", "BT709", "Full"] and head[7] == "1"
    assert frames == [["rgb8", str(53 * 3), "53"]] and np.array_equal(data.reshape(37, 53, 3), a8)
    a16 = (rng.random((19, 23, 3)) * 65535).astype(np.uint16)
    for inter in (False, True):
        p = str(tmp_path / f"b{int(inter)}.png")
        open(p, "wb").write(png_bytes(a16, inter))
        head, frames, data = read_dump(helper, p, str(tmp_path / "b.bin"))
        assert frames == [["rgb16", str(23 * 6), "23"]] and np.array_equal(data.view(np.uint16).reshape(19, 23, 3), a16), inter
    p = str(tmp_path / "c.png")
    open(p, "wb").write(png_bytes(a8, True))
    assert np.array_equal(read_dump(helper, p, str(tmp_path / "c.bin"))[2].reshape(37, 53, 3), a8)
    # like the reference, only RGB layouts are accepted (img.rs:17-37 is todo!() for the rest)
    p = str(tmp_path / "g.png")
    Image.fromarray(a8[..., 0]).save(p)
    assert "not implemented" in run(helper, "source", p, str(tmp_path / "g.bin"))
    p = str(tmp_path / "rgba.png")
    Image.fromarray(np.dstack([a8, a8[..., :1]])).save(p)
    assert "not implemented" in run(helper, "source", p, str(tmp_path / "rgba.bin"))
    # fewer than PROBE_LEN bytes: io::ErrorKind::UnexpectedEof (input_image.rs:51-53)
    p = str(tmp_path / "short.png")
    open(p, "wb").write(b"\x89PNG\r\n\x1a\n")
    assert "unexpected end of file" in run(helper, "source", p, str(tmp_path / "s.bin"))
    p = str(tmp_path / "x.jpg")
    open(p, "wb").write(b"\xff\xd8\xff\xe0" + bytes(200))
    assert "detected as JPEG but no decoder is available" in run(helper, "source", p, str(tmp_path / "j.bin"))


@pytest.mark.parametrize("bits16", [False, True])
def test_animated_png_yields_every_frame_composed_onto_the_canvas(helper, tmp_path, bits16):
    """every frame of a decoded image becomes a frame of the stream (input_image.rs:115-128): an animated PNG (acTL / fcTL / fdAT) --
    sub-rectangle frames, all three dispose operations -- against Pillow's own APNG decoder; a plain PNG still has one frame"""
    from PIL import Image
    rng = np.random.default_rng(3)
    w, h, n = 61, 47, 6
    base = (rng.random((h, w, 3)) * 255).astype(np.uint8)
    frames = [base.copy()]
    for k in range(1, n):  # local changes: Pillow stores the bounding box of the difference only
        f = frames[-1].copy()
        y0, x0 = 3 + 5 * k, 2 + 7 * k
        f[y0:y0 + 9, x0:x0 + 11] = (rng.random((9, 11, 3)) * 255).astype(np.uint8)
        frames.append(f)
    p = str(tmp_path / "a.png")
    for disposal in (0, 1, 2, [0, 1, 2, 1, 0, 2]):
        Image.fromarray(frames[0]).save(p, save_all=True, append_images=[Image.fromarray(f) for f in frames[1:]], disposal=disposal, blend=0)
        im = Image.open(p)
        assert getattr(im, "n_frames", 1) == n
        want = []
        for k in range(n):
            im.seek(k)
            want.append(np.asarray(im.convert("RGB")).copy())
        head, fr, data = read_dump(helper, p, str(tmp_path / "a.bin"))
        assert head[0] == "PNG/turbo-metrics-hip" and head[7] == str(n) and len(fr) == n, (head, len(fr))
        got = data.reshape(n, h, w, 3)
        for k in range(n):
            assert np.array_equal(got[k], want[k]), (disposal, k)
        assert len(read_dump(helper, p, str(tmp_path / "b.bin"), "--skip", 4)[1]) == n - 4
    if bits16:  # 16-bit samples: a hand-made APNG (Pillow writes 8-bit RGB only) -- IHDR, acTL, fcTL, IDAT, fcTL, fdAT, IEND
        import struct, zlib
        a = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
        patch = rng.integers(0, 65536, (8, 10, 3), dtype=np.uint16)

        def chunk(t, body):
            return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))

        def zimg(arr):
            return zlib.compress(b"".join(b"\x00" + row.astype(">u2").tobytes() for row in arr))
        blob = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) + chunk(b"acTL", struct.pack(">II", 2, 0))
                + chunk(b"fcTL", struct.pack(">IIIIIHHBB", 0, w, h, 0, 0, 1, 10, 0, 0)) + chunk(b"IDAT", zimg(a))
                + chunk(b"fcTL", struct.pack(">IIIIIHHBB", 1, 10, 8, 5, 7, 1, 10, 0, 0)) + chunk(b"fdAT", struct.pack(">I", 2) + zimg(patch)) + chunk(b"IEND", b""))
        open(p, "wb").write(blob)
        head, fr, data = read_dump(helper, p, str(tmp_path / "c.bin"))
        assert len(fr) == 2 and fr[0][0] == "rgb16"
        got = data.view(np.uint16).reshape(2, h, w, 3)
        b = a.copy(); b[7:15, 5:15] = patch
        assert np.array_equal(got[0], a) and np.array_equal(got[1], b)
        # a frame that sticks out of the canvas, an fdAT without its fcTL: errors, not crashes
        bad = blob.replace(struct.pack(">IIIIIHHBB", 1, 10, 8, 5, 7, 1, 10, 0, 0), struct.pack(">IIIIIHHBB", 1, 10, 8, 55, 7, 1, 10, 0, 0))
        open(p, "wb").write(bad)
        assert "outside the canvas" in run(helper, "source", p, str(tmp_path / "d.bin"))


def test_ppm_and_pfm_sources(helper, tmp_path):
    rng = np.random.default_rng(2)
    a8 = (rng.random((9, 31, 3)) * 255).astype(np.uint8)
    p = str(tmp_path / "a.ppm")
    open(p, "wb").write(b"P6\n# comment\n31 9\n255\n" + a8.tobytes() + bytes(64))
    _, frames, data = read_dump(helper, p, str(tmp_path / "a.bin"))
    assert frames[0][0] == "rgb8" and np.array_equal(data.reshape(9, 31, 3), a8)
    a16 = (rng.random((9, 31, 3)) * 65535).astype(np.uint16)
    open(p, "wb").write(b"P6 31 9 65535\n" + a16.astype(">u2").tobytes())
    _, frames, data = read_dump(helper, p, str(tmp_path / "a.bin"))
    assert frames[0][0] == "rgb16" and np.array_equal(data.view(np.uint16).reshape(9, 31, 3), a16)
    af = rng.random((9, 31, 3)).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    open(p, "wb").write(b"PF\n31 9\n-1.0\n" + af[::-1].tobytes())  # PFM rows run bottom to top
    _, frames, data = read_dump(helper, p, str(tmp_path / "f.bin"))
    assert frames[0][0] == "rgbf32" and np.array_equal(data.view(np.float32).reshape(9, 31, 3), af)


def write_y4m(path, frames, w, h, bits, extra=""):
    cs = "C420jpeg" if bits == 8 else f"C420p{bits}"
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F30:1 Ip A1:1 {cs}{extra}\n".encode())
        for planes in frames:
            f.write(b"FRAME\n")
            for pl in planes:
                f.write(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes())


@pytest.mark.parametrize("w,h,bits", [(70, 38, 8), (33, 67, 8), (46, 30, 10), (322, 271, 8), (318, 258, 10)])  # >= 256 rows: the workers split the picture
def test_y4m_pictures_reach_the_engine_as_planar_frames(helper, tmp_path, w, h, bits):
    """planar 4:2:0 streams are handed over as they are (HwFrame::Planar420 -> tm_engine_set_frame_i420): the frame's three
    planes are the stream's bytes; the GPU tier checks that the engine converts them exactly like the repacked NV12 / P016 surface"""
    pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(3)]
    p = str(tmp_path / "v.y4m")
    write_y4m(p, [pr[0] for pr in pairs], w, h, bits)
    head, frames, data = read_dump(helper, p, str(tmp_path / "v.bin"))
    assert head[0] == "Y4M/I420" + ("" if bits == 8 else "p10") + "/turbo-metrics-hip" and head[1:3] == [str(w), str(h)]
    want_mc = "BT601_525" if h <= 525 else "BT709"
    assert head[3:7] == [want_mc, want_mc, "BT709", "Limited"] and head[7] == "3"
    assert len(frames) == 3
    bps, cw, ch = (1 if bits == 8 else 2), (w + 1) // 2, (h + 1) // 2
    assert frames[0] == ["i420", str(bits), str(w * bps), str(cw * bps), str(h), str(ch)]
    per = (w * h + 2 * cw * ch) * bps
    for n, pr in enumerate(pairs):
        want = b"".join(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes() for pl in pr[0])
        assert data[n * per:(n + 1) * per].tobytes() == want
    # --skip drops leading pictures; XCOLORRANGE=FULL is carried through (and refused by the engine later, like todo!())
    head, frames, _ = read_dump(helper, p, str(tmp_path / "v2.bin"), "--skip", 2)
    assert len(frames) == 1
    write_y4m(p, [pairs[0][0]], w, h, bits, " XCOLORRANGE=FULL")
    assert read_dump(helper, p, str(tmp_path / "v3.bin"))[0][6] == "Full"
    # headerless planar stream with the geometry from the command line
    raw = str(tmp_path / "v.yuv")
    with open(raw, "wb") as f:
        for pr in pairs:
            for pl in pr[0]:
                f.write(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes())
    head, frames, data2 = read_dump(helper, raw, str(tmp_path / "r.bin"), "--width", w, "--height", h, "--bits", bits, "--cp", 1, "--mc", 1, "--tc", 1)
    assert head[3:6] == ["BT709", "BT709", "BT709"] and len(frames) == 3 and np.array_equal(data2, data)
    assert "not a PNG" in run(helper, "source", raw, str(tmp_path / "n.bin"))
    # the same streams through a pipe (stdin): the probe bytes are handed back to the source, nothing is buffered whole
    for path, extra in ((p, ()), (raw, ("--width", w, "--height", h, "--bits", bits))):
        outp = str(tmp_path / "pipe.bin")
        if path == p:
            write_y4m(p, [pr[0] for pr in pairs], w, h, bits)
        r = subprocess.run([helper, "source", "-", outp] + [str(a) for a in extra], stdin=open(path, "rb"), capture_output=True, text=True, env=dict(os.environ, TM_PACK10=PACK10))
        assert r.stdout.strip().split("\n")[0].split()[1:3] == [str(w), str(h)] and len(r.stdout.strip().split("\n")) == 4, r.stdout
        assert np.array_equal(np.fromfile(outp, np.uint8), data)


@pytest.mark.parametrize("w,h,bits,frames", [(322, 271, 8, 41), (1920, 1080, 10, 9)])
def test_read_ahead_delivers_the_stream_in_order_through_a_small_ring(helper, tmp_path, w, h, bits, frames):
    """regular files are read AHEAD of the next_frame calls by a pool of readers (pieces of 2 MB: the 1080p 10-bit picture is three)
    into a ring of lookahead + 1 + ahead surfaces: more pictures than ring slots, every picture's bytes in stream order, the same
    as the synchronous reader; a stream cut in the middle of picture N delivers N pictures and then fails like the synchronous one"""
    rng = np.random.default_rng(7)
    bps, cw, ch = (1 if bits == 8 else 2), (w + 1) // 2, (h + 1) // 2
    pics = []
    for n in range(frames):
        hi = 256 if bits == 8 else 1024
        pics.append([rng.integers(0, hi, (h, w), dtype=np.uint16), rng.integers(0, hi, (ch, cw), dtype=np.uint16), rng.integers(0, hi, (ch, cw), dtype=np.uint16)])
    p = str(tmp_path / "v.y4m")
    write_y4m(p, pics, w, h, bits)
    want = b"".join(b"".join(pl.astype(np.uint8 if bits == 8 else "<u2").tobytes() for pl in pr) for pr in pics)
    for extra in (("--readahead", 1), ("--readahead", 0), ("--readahead", 1, "--lookahead", 4), ("--readahead", 1, "--skip", 3)):
        head, fr, data = read_dump(helper, p, str(tmp_path / "o.bin"), *extra)
        skip = 3 if "--skip" in extra else 0
        assert len(fr) == frames - skip and data.tobytes() == want[skip * len(want) // frames:], extra
    per = len(want) // frames
    cut = str(tmp_path / "cut.y4m")
    blob = open(p, "rb").read()
    open(cut, "wb").write(blob[: len(blob) - per // 2])  # the last picture is incomplete
    outs = [run(helper, "source", cut, str(tmp_path / f"c{i}.bin"), "--readahead", i) for i in (0, 1)]
    for o in outs:
        assert o.strip().split("\n")[-1].startswith("ERROR: truncated picture") and len(o.strip().split("\n")) == 1 + (frames - 1) + 1, o[-300:]
    assert np.array_equal(np.fromfile(str(tmp_path / "c0.bin"), np.uint8), np.fromfile(str(tmp_path / "c1.bin"), np.uint8))


@pytest.mark.parametrize("w,h,frames", [(46, 30, 3), (318, 258, 5), (1920, 1080, 7), (770, 300, 4)])
def test_10_bit_pictures_are_packed_three_samples_to_a_word_on_their_way_into_the_ring(helper, tmp_path, w, h, frames, monkeypatch):
    """round 6: a yuv420p10 stream reaches the engine as HwFrame::Planar420P10 -- the readers pack every row (tm_p10_pack_rows) while they
    copy the picture into the page-locked ring: through the read-ahead pool (pieces of whole rows), the synchronous reader with its row
    workers, and a pipe; unpacked by the numpy statement of the layout the samples are the stream's"""
    monkeypatch.setattr(sys.modules[__name__], "PACK10", "1")
    rng = np.random.default_rng(11)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    pics = [[rng.integers(0, 1024, (h, w), dtype=np.uint16), rng.integers(0, 1024, (ch, cw), dtype=np.uint16), rng.integers(0, 1024, (ch, cw), dtype=np.uint16)] for _ in range(frames)]
    p = str(tmp_path / "v.y4m")
    write_y4m(p, pics, w, h, 10)
    wy, wc = tm.synth.p10_row_words(w), tm.synth.p10_row_words(cw)
    per = (h * wy + 2 * ch * wc) * 4

    def check(fr, data, skip=0):
        assert len(fr) == frames - skip and all(f == ["p10", "10", str(wy * 4), str(wc * 4), str(h), str(ch)] for f in fr)
        words = data.view(np.uint32)
        for n in range(skip, frames):
            pic = words[(n - skip) * per // 4:(n - skip + 1) * per // 4]
            got = (tm.synth.p10_unpack_plane(pic[:h * wy].reshape(h, wy), w), tm.synth.p10_unpack_plane(pic[h * wy:h * wy + ch * wc].reshape(ch, wc), cw),
                   tm.synth.p10_unpack_plane(pic[h * wy + ch * wc:].reshape(ch, wc), cw))
            assert all(np.array_equal(g, pl) for g, pl in zip(got, pics[n])), n
            assert np.array_equal(pic[:h * wy].reshape(h, wy), tm.synth.p10_pack_plane(pics[n][0]))  # absent samples and the two top bits are 0
    for extra in (("--readahead", 1), ("--readahead", 0), ("--readahead", 1, "--lookahead", 4, "--skip", 2)):
        head, fr, data = read_dump(helper, p, str(tmp_path / "o.bin"), *extra)
        assert head[0] == "Y4M/I420p10/turbo-metrics-hip"
        check(fr, data, 2 if "--skip" in extra else 0)
    outp = str(tmp_path / "pipe.bin")
    r = subprocess.run([helper, "source", "-", outp], stdin=open(p, "rb"), capture_output=True, text=True, env=dict(os.environ, TM_PACK10="1"))
    check([t.split() for t in r.stdout.strip().split("\n")[1:]], np.fromfile(outp, np.uint8))
    # 12-bit streams stay 16-bit words
    write_y4m(p, [[pl * 4 for pl in pics[0]]], w, h, 12)
    assert read_dump(helper, p, str(tmp_path / "o12.bin"))[1][0][0] == "i420"


def test_y4m_rejects_what_the_reference_cannot_represent(helper, tmp_path):
    p = str(tmp_path / "v.y4m")
    open(p, "wb").write(b"YUV4MPEG2 W64 H64 F30:1 Ip A1:1 C444\nFRAME\n" + bytes(64 * 64 * 3))
    assert "not implemented" in run(helper, "source", p, str(tmp_path / "o.bin"))


def test_corrupt_headers_end_in_errors_not_allocations(helper, tmp_path):
    """found by tools/fuzz_sources.py: a PNG whose IHDR promises 21 x 889 192 461 pixels asked for a 57 GB buffer"""
    import struct, zlib
    good = png_bytes(np.zeros((8, 8, 3), np.uint8))
    ihdr = bytearray(good[16:29])
    for w, h in ((21, 889192461), (70000, 8), (8, 70000)):
        ihdr[0:8] = struct.pack(">II", w, h)
        bad = good[:16] + bytes(ihdr) + struct.pack(">I", zlib.crc32(b"IHDR" + bytes(ihdr))) + good[33:]
        p = str(tmp_path / "bad.png")
        open(p, "wb").write(bad)
        assert run(helper, "source", p, str(tmp_path / "o.bin")).startswith("ERROR")
    # a plausible size that the 60 compressed bytes cannot possibly inflate to
    ihdr[0:8] = struct.pack(">II", 4000, 4000)
    open(p, "wb").write(good[:16] + bytes(ihdr) + struct.pack(">I", zlib.crc32(b"IHDR" + bytes(ihdr))) + good[33:])
    assert "corrupt" in run(helper, "source", p, str(tmp_path / "o.bin"))
    for blob in (b"P6\n99999999 3\n255\n" + bytes(64), b"PF\n4000000000 4000000000\n-1.0\n" + bytes(64)):
        q = str(tmp_path / ("bad.ppm" if blob[1:2] == b"6" else "bad.pfm"))
        open(q, "wb").write(blob)
        assert run(helper, "source", q, str(tmp_path / "o.bin")).startswith("ERROR")
    y = str(tmp_path / "bad.y4m")
    open(y, "wb").write(b"YUV4MPEG2 W100000 H64 F30:1 Ip A1:1 C420jpeg\nFRAME\n" + bytes(256))
    assert run(helper, "source", y, str(tmp_path / "o.bin")).startswith("ERROR")


# ---- `--ranks N` without a GPU: launcher, pipes, the one reduce --------------------------------------------------------
def test_rank_launcher_and_pipe_reduce_bring_every_score_to_rank_zero(helper):
    """host/ranks.cpp through the helper: N processes started by the launcher, contiguous blocks of the decode indices (the blocks of
    shard.py::shard_range), one reduce over the launcher's pipes, rank 0 alone on stdout, every value exactly the one its rank produced."""
    from tm_pkg import tm as _tm
    for total, world in ((23, 1), (23, 2), (23, 3), (7, 8), (100, 5)):
        got = [tuple(map(int, l.split())) for l in run(helper, "shard", total, world).strip().split("\n")]
        assert got == [_tm.shard.shard_range(total, r, world) for r in range(world)]
    for world, total, every in ((1, 9, 0), (2, 23, 0), (3, 23, 4), (5, 3, 0), (4, 40, 3)):
        r = subprocess.run([helper, "ranks", str(world), str(total), str(every)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        lo, hi = _tm.shard.shard_range(total, 0, world)
        assert lines[0] == f"lo {lo} hi {hi} decoded {total} pipe"
        sel = [dc for dc in range(total) if not (every > 1 and dc != 0 and dc % every != 0)]
        want = [(float("inf") if dc == 5 else 30.0 + 0.1 * dc, 100.0 / (1.0 + dc) - 7.0) for dc in sel]
        got = [tuple(float.fromhex(t) for t in l.split()) for l in lines[1:]]
        assert got == want  # bit for bit: every score was added to zeros only
        assert "must not be heard" not in r.stdout


def test_rank_launcher_ends_the_job_when_a_rank_fails(helper):
    """launch.py's rules in C++: the first failing rank's exit code is the launcher's, and the ranks still running -- one of them
    blocked for good -- are terminated by PID instead of being waited for."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([helper, "ranks", "4", "23", "0", "2", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "rank 2 of 4 failed (exit code 3)" in r.stderr
    assert time.monotonic() - t0 < 30
    # rank 0 gone: the others' writes fail instead of blocking
    r = subprocess.run([helper, "ranks", "3", "200000", "0", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    # a signal to the launcher stops its ranks (one of them blocked for good): nothing is left behind
    import signal
    p = subprocess.Popen([helper, "ranks", "3", "23", "0", "-1", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, TM_TEST_PRINT_PID="1"))
    assert p.stdout.readline().startswith("launcher ")
    time.sleep(0.5)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) >= 1  # (the rank that hangs is still there; the others may have finished or wait in the reduce)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    assert p.returncode == 128 + signal.SIGTERM and "stopping the ranks" in err
    time.sleep(0.2)
    assert not [k for k in kids if os.path.exists(f"/proc/{k}") and "Z" not in open(f"/proc/{k}/stat").read().split()[2]]


# ---- the command line on the device ------------------------------------------------------------------------------------
def cli(*args, stdin=None, env=None):
    r = subprocess.run([CLI] + [str(a) for a in args], input=stdin, capture_output=True, check=False, env=None if env is None else dict(os.environ, **env))
    return r.returncode, r.stdout.decode(), r.stderr.decode()


@pytest.mark.gpu
def test_cli_png_pair_all_outputs(tmp_path):
    from PIL import Image
    from oracle import oracle as O
    w, h = 160, 96
    r8, d8 = tm.synth.rgb8_pair(w, h)
    pr, pd = str(tmp_path / "r.png"), str(tmp_path / "d.png")
    Image.fromarray(r8).save(pr)
    Image.fromarray(d8).save(pd)
    want, _ = O.ssimulacra2_from_linear(O.rgb8_to_linear(r8), O.rgb8_to_linear(d8))
    _, want_psnr = O.psnr(O.rgb8_to_linear(r8), O.rgb8_to_linear(d8))
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json")
    assert rc == 0, err
    d = json.loads(out)
    assert d["frame_count"] == 1 and abs(d["ssimulacra2"]["scores"][0] - want) <= 1e-9 and d["psnr"]["scores"][0] == want_psnr
    assert list(d) == ["frame_count", "psnr", "ssimulacra2"]
    assert "Processed: 1 (decoded: ~1) frame pairs" in err and "codec=PNG/" in err
    score = d["ssimulacra2"]["scores"][0]
    rc, out, _ = cli(pr, pd, "--metrics", "ssimulacra2", "--output", "csv")
    assert rc == 0 and out == f"ssimulacra2\n{rust_display(score)}\nssimulacra2\n{rust_display(score)}\n"
    rc, out, _ = cli(pr, pd, "--metrics=ssimulacra2", "--output=json-lines")
    lines = out.strip().split("\n")
    assert json.loads(lines[0]) == {"ssimulacra2": score} and json.loads(lines[1])["ssimulacra2"]["p50"] == score
    rc, out, _ = cli(pr, pd, "-m", "ssimulacra2")
    assert rc == 0 and out.startswith("SSIMULACRA2: Stats {\n    min: ") and out.count("\n") == 14
    # identical images: PSNR = inf (CSV "inf", JSON null)
    rc, out, _ = cli(pr, pr, "-m", "psnr", "--output", "csv")
    assert out.split("\n")[1] == "inf"
    rc, out, _ = cli(pr, pr, "-m", "psnr", "--output", "json-lines")
    assert out.split("\n")[0] == '{"psnr":null}'
    # stdin as one of the inputs
    rc, out, _ = cli("-", pd, "-m", "ssimulacra2", "--output", "csv", stdin=open(pr, "rb").read())
    assert rc == 0 and out.split("\n")[1] == rust_display(score)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [8, 10])
def test_cli_y4m_stream_frame_selection_batching_and_pipeline(tmp_path, bits):
    from oracle import oracle as O
    w, h, n = 96, 64, 11
    pairs = [tm.synth.yuv420_pair(w, h, i, bits) for i in range(n)]
    pr, pd = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    write_y4m(pr, [p[0] for p in pairs], w, h, bits)
    write_y4m(pd, [p[1] for p in pairs], w, h, bits)
    want = []
    for p in pairs:
        sr, pit, ch = tm.synth.pack_biplanar(p[0], w, h, bits)
        sd, _, _ = tm.synth.pack_biplanar(p[1], w, h, bits)
        m = 1  # height 64 <= 525 -> BT601_525 by the reference's fallback (color.rs:51-78)
        lr = O.yuv420_biplanar_to_linear(sr, pit, ch, w, h, 8 if bits == 8 else 16, m)
        ld = O.yuv420_biplanar_to_linear(sd, pit, ch, w, h, 8 if bits == 8 else 16, m)
        want.append(O.ssimulacra2_from_linear(lr, ld)[0])

    def scores(*extra):
        rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "--output", "json", *extra)
        assert rc == 0, err
        return json.loads(out)["ssimulacra2"]["scores"], err
    base, err = scores("--batch", 4)
    assert len(base) == n and max(abs(a - b) for a, b in zip(base, want)) <= 1e-9
    assert "mc=BT601_525" in err and f"frame_count={n}" in err
    assert scores("--batch", 1, "--no-pipeline")[0] == base      # batching and pipelining never change a bit
    assert scores("--batch", 3)[0] == base and scores("--batch", 16, "--full-sums")[0] == base
    assert scores("--every", 3)[0] == [base[i] for i in (0, 3, 6, 9)]
    assert scores("--skip", 2, "--frames", 3)[0] == base[2:5]
    got, _ = scores("--skip-ref", 1)
    assert len(got) == n - 1 and got != base[1:]                 # reference frame i+1 against distorted frame i
    # the reference's own loop (one blocking compute_one per pair) and the same loop on compute_one_deferred + collect
    for mode in ("reference", "deferred"):
        assert scores("--loop", mode)[0] == base
        assert scores("--loop", mode, "--every", 3)[0] == [base[i] for i in (0, 3, 6, 9)]
        assert scores("--loop", mode, "--skip", 2, "--frames", 3)[0] == base[2:5]
    rc, out1, _ = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines")
    rc2, out2, _ = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--loop", "deferred")
    assert rc == 0 and rc2 == 0 and out1 == out2                 # byte-identical stdout
    for depth in (3, 4, 8):                                      # more pairs in flight than the clip has frames, too
        rc2, out2, err2 = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--loop", "deferred", "--in-flight", depth)
        assert rc2 == 0 and out2 == out1, (depth, err2)
        assert scores("--loop", "deferred", "--in-flight", depth, "--every", 3)[0] == [base[i] for i in (0, 3, 6, 9)]
    # explicit BT.709 metadata overrides the fallback
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "--output", "json", "--color-primaries", 1, "--matrix-coefficients", 1, "--transfer-characteristics", 1)
    assert rc == 0 and "mc=BT709" in err and json.loads(out)["ssimulacra2"]["scores"] != base


@pytest.mark.gpu
def test_cli_shards_the_stream_over_devices_with_identical_output(tmp_path):
    """--devices N: contiguous shards of the decode indices, one host thread + engines per shard, scores concatenated in shard
    order -- stdout is byte-identical to the single-device run for every format and frame selection.  On the 1-GPU test box the
    shards share the device (TM_SHARE_DEVICE=1)."""
    w, h, n, bits = 96, 64, 23, 8
    pairs = [tm.synth.yuv420_pair(w, h, i, bits) for i in range(n)]
    pr, pd = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    write_y4m(pr, [p[0] for p in pairs], w, h, bits)
    write_y4m(pd, [p[1] for p in pairs], w, h, bits)
    share = {"TM_SHARE_DEVICE": "1"}
    for extra in ((), ("--every", 4), ("--skip", 3, "--frames", 11), ("--skip-ref", 2, "--every", 3, "--frames", 17), ("-m", "psnr", "--batch", 2)):
        for fmt in ("json", "json-lines", "csv", "default"):
            one = cli(pr, pd, "-m", "ssimulacra2", "--output", fmt, "--batch", 4, *extra)
            assert one[0] == 0, one[2]
            for ndev in (2, 3, 5):
                many = cli(pr, pd, "-m", "ssimulacra2", "--output", fmt, "--batch", 4, "--devices", ndev, *extra, env=share)
                assert many[0] == 0, many[2]
                assert many[1] == one[1], (extra, fmt, ndev)
                assert "on %d devices" % ndev in many[2] or n < ndev
    import torch
    rc, _, err = cli(pr, pd, "-m", "ssimulacra2", "--devices", torch.cuda.device_count() + 1)
    assert rc == 1 and "GPU(s) visible" in err
    # a pipe has no length: one device, with a warning
    rc, out, err = cli("-", pd, "-m", "ssimulacra2", "--output", "csv", "--devices", 2, stdin=open(pr, "rb").read(), env=share)
    assert rc == 0 and "running on one device" in err and out == cli(pr, pd, "-m", "ssimulacra2", "--output", "csv")[1]


@pytest.mark.gpu
def test_cli_ranks_one_process_per_device_and_one_reduce_with_identical_output(tmp_path):
    """--ranks N: the launcher starts N rank processes before any GPU call, rank r scores shard_range(r), ONE
    reduce(sum, f64) of the zero-padded score vector brings everything to rank 0, whose stdout is byte-identical with the one-device
    run.  On the 1-GPU test box: N ranks sharing the device over the pipe transport (RCCL refuses two ranks on one GPU), and a ONE-rank
    communicator over RCCL itself (libturbometrics_rccl.so: ncclCommInitRank + ncclReduce on the device)."""
    w, h, n, bits = 96, 64, 23, 8
    pairs = [tm.synth.yuv420_pair(w, h, i, bits) for i in range(n)]
    pr, pd = str(tmp_path / "r.y4m"), str(tmp_path / "d.y4m")
    write_y4m(pr, [p[0] for p in pairs], w, h, bits)
    write_y4m(pd, [p[1] for p in pairs], w, h, bits)
    pipes = {"TM_SHARE_DEVICE": "1", "TM_RANK_TRANSPORT": "pipe", "TM_RANK_TIMEOUT_S": "300"}
    for extra in ((), ("--every", 4), ("--skip", 3, "--frames", 11), ("--skip-ref", 2, "--every", 3, "--frames", 17), ("-m", "psnr", "-m", "ssim", "--batch", 2)):
        for fmt in ("json", "json-lines", "csv", "default"):
            one = cli(pr, pd, "-m", "ssimulacra2", "--output", fmt, "--batch", 4, *extra)
            assert one[0] == 0, one[2]
            for nr in (2, 3) if fmt in ("json", "csv") else (2,):
                many = cli(pr, pd, "-m", "ssimulacra2", "--output", fmt, "--batch", 4, "--ranks", nr, *extra, env=pipes)
                assert many[0] == 0, many[2]
                assert many[1] == one[1], (extra, fmt, nr)
                assert "on %d ranks (pipe)" % nr in many[2]
    one = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--batch", 4)
    # more ranks than frames: the tail ranks hold empty blocks
    many = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--batch", 4, "--frames", 3, "--ranks", 5, env=pipes)
    assert many[0] == 0 and many[1] == cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--batch", 4, "--frames", 3)[1]
    # RCCL: a one-rank communicator on the device (everything but the second GPU)
    rccl = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "--output", "json-lines", "--batch", 4, "--ranks", 1, env={"TM_RANK_TRANSPORT": "rccl", "TM_RANK_TIMEOUT_S": "300"})
    assert rccl[0] == 0, rccl[2]
    assert rccl[1] == one[1] and "on 1 ranks (rccl)" in rccl[2]
    # no silent fallback to pipes when the collective library cannot be loaded
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "--ranks", 1, env={"TM_RANK_TRANSPORT": "rccl", "TM_RCCL_LIB": str(tmp_path / "nowhere.so"), "TM_RANK_TIMEOUT_S": "120"})
    assert rc != 0 and "needs libturbometrics_rccl.so" in err and out == ""
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "--ranks", 1, env={"TM_RANK_TRANSPORT": "carrier-pigeon", "TM_RANK_TIMEOUT_S": "120"})
    assert rc != 0 and "possible values: rccl, pipe" in err
    # without sharing, more ranks than GPUs is an error of every rank and of the launcher
    import torch
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "--ranks", torch.cuda.device_count() + 1, env={"TM_RANK_TRANSPORT": "pipe", "TM_RANK_TIMEOUT_S": "120"})
    assert rc != 0 and "GPU(s) visible" in err and out == ""
    # a failing rank ends the job: an unreadable distorted file
    rc, out, err = cli(pr, str(tmp_path / "missing.y4m"), "-m", "ssimulacra2", "--ranks", 2, env=pipes)
    assert rc != 0 and "failed" in err
    rc, _, err = cli("-", pd, "-m", "ssimulacra2", "--ranks", 2, stdin=open(pr, "rb").read(), env=pipes)
    assert rc == 1 and "not stdin" in err
    rc, _, err = cli(pr, pd, "-m", "ssimulacra2", "--ranks", 2, "--devices", 2, env=pipes)
    assert rc == 1 and "exclude each other" in err


@pytest.mark.gpu
def test_cli_error_paths(tmp_path):
    from PIL import Image
    a = (np.random.default_rng(0).random((32, 48, 3)) * 255).astype(np.uint8)
    p1, p2 = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(a).save(p1)
    Image.fromarray(a[:, :40]).save(p2)
    rc, _, err = cli(p1, p2, "-m", "ssimulacra2")
    assert rc == 1 and "Reference and distorted are not the same size" in err
    rc, _, err = cli("-", "-", "-m", "ssimulacra2")
    assert rc == 1 and "Can't read both reference and distorted from stdin" in err
    rc, _, err = cli(p1, str(tmp_path / "missing.png"), "-m", "ssimulacra2")
    assert rc == 1 and "Could not read distorted" in err
    rc, _, err = cli(p1, p1, "-m", "msssim")
    assert rc == 1 and "Could not initialize engine" in err     # 48x32: no fifth dyadic scale for MS-SSIM
    rc, out, err = cli(p1, p1, "-m", "ssim", "-m", "psnr", "--output", "csv")
    assert rc == 0 and out.split("\n")[:2] == ["psnr,ssim", "inf,1"]
    rc, _, err = cli(p1, p1, "-m", "vmaf")
    assert rc == 2 and "possible values: psnr, ssim, msssim, ssimulacra2" in err
    rc, _, err = cli(p1)
    assert rc == 2
    for bad in ("1", "9", "x"):
        rc, _, err = cli(p1, p1, "-m", "ssimulacra2", "--loop", "deferred", "--in-flight", bad)
        assert rc == 2 and "--in-flight <N>" in err
    rc, out, err = cli(p1, p1, "-m", "ssim", "--output", "csv", "--in-flight", 4)
    assert rc == 0 and "--in-flight belongs to --loop deferred" in err and out.split("\n")[:2] == ["ssim", "1"]
    y = str(tmp_path / "f.y4m")
    write_y4m(y, [tm.synth.yuv420_pair(48, 32, 0, 8)[0]], 48, 32, 8, " XCOLORRANGE=FULL")
    rc, _, err = cli(y, y, "-m", "ssimulacra2")
    assert rc == 1 and "unsupported" in err                      # full-range YUV: todo!() in the reference


@pytest.mark.gpu
def test_cli_baseline_config_1_full_hd_png_pair(tmp_path):
    """BASELINE.json configs[0]: one 1080p PNG pair.  The CLI's score equals the committed golden (oracle, GPU arithmetic) to
    1e-4 and lies within the reference's own 0.25 band of its CPU path (examples/cpu.rs restated, examples/compare.rs:72)."""
    from PIL import Image
    case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "scores_regression.json")))["cases"]
            if c["kind"] == "rgb8" and c["width"] == 1920][0]
    r8, d8 = tm.synth.rgb8_pair(1920, 1080)
    pr, pd = str(tmp_path / "r.png"), str(tmp_path / "d.png")
    Image.fromarray(r8).save(pr, compress_level=1)
    Image.fromarray(d8).save(pd, compress_level=1)
    rc, out, err = cli(pr, pd, "-m", "ssimulacra2", "-m", "psnr", "-m", "ssim", "-m", "msssim", "--output", "json")
    assert rc == 0, err
    d = json.loads(out)
    assert abs(d["ssimulacra2"]["scores"][0] - case["ssimulacra2"]) <= 1e-4
    assert abs(d["ssimulacra2"]["scores"][0] - case["cpu_path_ssimulacra2"]) < 0.25
    assert d["psnr"]["scores"][0] == case["psnr"]
    assert abs(d["ssim"]["scores"][0] - case["ssim"]) <= 1e-6 and abs(d["msssim"]["scores"][0] - case["msssim"]) <= 1e-6
