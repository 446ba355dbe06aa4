"""Compressed-video input: IVF / Matroska demuxing, H.264 / AV1 / MPEG-2 sequence-header parsing and the decoder
pipe of turbo-metrics_amd/host/video_input.cpp, on files built here bit by bit (writers below) -- the reference holds no video
fixture.  The decoder is an external process on this platform; the tests use a stand-in that records the elementary stream it
is fed and answers with a prepared YUV4MPEG2 stream."""
import os
import stat
import struct
import subprocess
import sys

import numpy as np
import pytest

from tm_pkg import tm
from tests.test_host_cli import helper, run, read_dump, write_y4m  # noqa: F401  (fixture + helpers)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- bit-level writers -------------------------------------------------------------------------------------------------
class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        return self.u(n - 1, 0).u(n, v)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def bytes(self, trailing=True):
        b = list(self.bits)
        if trailing:
            b.append(1)
        while len(b) % 8:
            b.append(0)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def escape_rbsp(rbsp):
    out, zeros = bytearray(), 0
    for x in rbsp:
        if zeros >= 2 and x <= 3:
            out.append(3)
            zeros = 0
        out.append(x)
        zeros = zeros + 1 if x == 0 else 0
    return bytes(out)


def h264_sps(w, h, profile=100, depth=8, vui=None, crop=True, scaling=False):
    """SPS NAL unit (header byte included).  vui: None or (full_range, cp, tc, mc)"""
    mbs_w, mbs_h = (w + 15) // 16, (h + 15) // 16
    b = BitWriter().u(8, profile).u(8, 0).u(8, 40).ue(0)
    if profile >= 100:
        b.ue(1).ue(depth - 8).ue(depth - 8).u(1, 0)
        b.u(1, 1 if scaling else 0)
        if scaling:
            for i in range(8):
                b.u(1, 1 if i == 0 else 0)
                if i == 0:
                    for j in range(16):
                        b.se(1 if j < 3 else 0)  # next = last + delta
    b.ue(0).ue(0).ue(4)          # log2_max_frame_num, poc type 0, log2_max_poc_lsb
    b.ue(4).u(1, 0)              # max_num_ref_frames, gaps
    b.ue(mbs_w - 1).ue(mbs_h - 1).u(1, 1).u(1, 1)  # frame_mbs_only, direct_8x8
    cr, cb_ = (mbs_w * 16 - w) // 2, (mbs_h * 16 - h) // 2
    if crop and (cr or cb_):
        b.u(1, 1).ue(0).ue(cr).ue(0).ue(cb_)
    else:
        b.u(1, 0)
    if vui is None:
        b.u(1, 0)
    else:
        full, cp, tc, mc = vui
        b.u(1, 1).u(1, 1).u(8, 255).u(16, 4).u(16, 3)  # aspect ratio: extended SAR
        b.u(1, 0)                                        # overscan
        b.u(1, 1).u(3, 5).u(1, full).u(1, 1).u(8, cp).u(8, tc).u(8, mc)
        b.u(1, 0).u(1, 0).u(1, 0).u(1, 0).u(1, 0).u(1, 0)  # chroma loc, timing, hrd x2, pic_struct, bitstream_restriction
    return bytes([0x67]) + escape_rbsp(b.bytes())


def leb128(v):
    out = bytearray()
    while True:
        out.append((v & 0x7F) | (0x80 if v > 0x7F else 0))
        v >>= 7
        if not v:
            return bytes(out)


def av1_sequence_header_obu(w, h, profile=0, high_bitdepth=0, color=None, full=0, timing=False):
    b = BitWriter().u(3, profile).u(1, 0).u(1, 0)
    b.u(1, 1 if timing else 0)
    if timing:
        b.u(32, 1).u(32, 30).u(1, 1).u(1, 1)  # display tick, time scale, equal_picture_interval, uvlc(0) = '1'
        b.u(1, 0)                              # decoder_model_info_present
    b.u(1, 0).u(5, 0)                          # initial_display_delay_present, operating_points_cnt_minus_1
    b.u(12, 0).u(5, 8).u(1, 0)                 # idc, seq_level_idx 8 (> 7 -> tier bit)
    wb, hb = (w - 1).bit_length(), (h - 1).bit_length()
    b.u(4, wb - 1).u(4, hb - 1).u(wb, w - 1).u(hb, h - 1)
    b.u(1, 0)                                  # frame_id_numbers_present
    b.u(1, 0).u(1, 1).u(1, 1)                  # 128x128, filter_intra, intra_edge
    b.u(4, 0b1111).u(1, 1).u(2, 0b11)          # interintra.., order_hint, jnt_comp + ref_frame_mvs
    b.u(1, 1)                                  # seq_choose_screen_content_tools -> SELECT (2)
    b.u(1, 1)                                  # seq_choose_integer_mv
    b.u(3, 6)                                  # order_hint_bits_minus_1
    b.u(3, 0b011)                              # superres, cdef, restoration
    b.u(1, high_bitdepth)
    if profile != 1:
        b.u(1, 0)                              # mono_chrome
    if color is None:
        b.u(1, 0)
    else:
        b.u(1, 1).u(8, color[0]).u(8, color[1]).u(8, color[2])  # cp, tc, mc
    b.u(1, full)                               # color_range (profile 0: 4:2:0)
    b.u(2, 0).u(1, 0).u(1, 0)                  # chroma_sample_position, separate_uv_delta_q, film_grain_params_present
    payload = b.bytes()
    return bytes([(1 << 3) | 2]) + leb128(len(payload)) + payload  # OBU_SEQUENCE_HEADER, has_size_field


def mpeg2_sequence(w, h, color=None):
    b = BitWriter().u(12, w & 0xFFF).u(12, h & 0xFFF).u(4, 1).u(4, 4).u(18, 1000).u(1, 1).u(10, 20).u(1, 0).u(1, 0).u(1, 0)
    out = b"\x00\x00\x01\xB3" + b.bytes(trailing=False)
    ext = BitWriter().u(4, 1).u(8, 0x48).u(1, 1).u(2, 1).u(2, w >> 12).u(2, h >> 12).u(12, 0).u(1, 1).u(8, 0).u(1, 0).u(2, 0).u(5, 0)
    out += b"\x00\x00\x01\xB5" + ext.bytes(trailing=False)
    if color is not None:
        d = BitWriter().u(4, 2).u(3, 1).u(1, 1).u(8, color[0]).u(8, color[1]).u(8, color[2]).u(14, w).u(1, 1).u(14, h)
        out += b"\x00\x00\x01\xB5" + d.bytes(trailing=False)
    return out


def seqhdr(helper, kind, data):
    return [int(x) for x in run(helper, "seqhdr", kind, data.hex()).split()]


def test_h264_sps_size_depth_and_colour_description(helper):
    assert seqhdr(helper, "h264", h264_sps(1920, 1080, vui=(0, 1, 1, 1))) == [1, 1920, 1080, 8, 1, 1, 1, 1, 0]
    assert seqhdr(helper, "h264", h264_sps(1920, 1080)) == [1, 1920, 1080, 8, 1, 2, 2, 2, 0]               # no VUI: unspecified
    assert seqhdr(helper, "h264", h264_sps(720, 576, profile=66, vui=(1, 5, 6, 5))) == [1, 720, 576, 8, 1, 5, 5, 6, 1]
    assert seqhdr(helper, "h264", h264_sps(3840, 2160, profile=110, depth=10, vui=(0, 9, 16, 9))) == [1, 3840, 2160, 10, 1, 9, 9, 16, 0]
    assert seqhdr(helper, "h264", h264_sps(1280, 718, scaling=True, vui=(0, 6, 6, 6))) == [1, 1280, 718, 8, 1, 6, 6, 6, 0]  # scaling lists skipped, cropped height
    assert seqhdr(helper, "h264", h264_sps(1920, 1080, crop=False)) == [1, 1920, 1088, 8, 1, 2, 2, 2, 0]
    # an SPS whose payload contains 00 00 0x sequences goes through emulation prevention
    sps = h264_sps(16 * 257, 16, vui=(0, 1, 1, 1))
    assert seqhdr(helper, "h264", sps)[:3] == [1, 16 * 257, 16]
    # truncated / wrong NAL type: invalid, never a crash
    assert seqhdr(helper, "h264", sps[:6])[0] == 0 and seqhdr(helper, "h264", b"\x68" + sps[1:])[0] == 0


def test_av1_sequence_header_and_mpeg2_sequence(helper):
    assert seqhdr(helper, "av1", av1_sequence_header_obu(1920, 1080, color=(1, 1, 1))) == [1, 1920, 1080, 8, 1, 1, 1, 1, 0]
    assert seqhdr(helper, "av1", av1_sequence_header_obu(3840, 2160, high_bitdepth=1, color=(9, 16, 9), full=1, timing=True)) == [1, 3840, 2160, 10, 1, 9, 9, 16, 1]
    assert seqhdr(helper, "av1", av1_sequence_header_obu(640, 360)) == [1, 640, 360, 8, 1, 2, 2, 2, 0]
    td = bytes([(2 << 3) | 2, 0])  # a temporal delimiter OBU before it
    assert seqhdr(helper, "av1", td + av1_sequence_header_obu(352, 288))[:3] == [1, 352, 288]
    assert seqhdr(helper, "av1", td)[0] == 0
    assert seqhdr(helper, "mpeg2", mpeg2_sequence(720, 480, color=(6, 6, 6))) == [1, 720, 480, 8, 1, 6, 6, 6, 0]
    assert seqhdr(helper, "mpeg2", mpeg2_sequence(1920, 1080)) == [1, 1920, 1080, 8, 1, 2, 2, 2, 0]
    assert seqhdr(helper, "mpeg2", b"\x00\x00\x01\xB8")[0] == 0


# ---- container writers ---------------------------------------------------------------------------------------------------
def ivf_file(fourcc, w, h, packets):
    out = b"DKIF" + struct.pack("<HH4sHHIIII", 0, 32, fourcc, w, h, 30, 1, len(packets), 0)
    for i, p in enumerate(packets):
        out += struct.pack("<IQ", len(p), i) + p
    return out


def ebml_id(i):
    return i.to_bytes((i.bit_length() + 7) // 8, "big")


def ebml_size(n, length=None):
    for L in range(1, 9):
        if length in (None, L) and n < (1 << (7 * L)) - 1:
            return ((1 << (7 * L)) | n).to_bytes(L, "big")
    raise ValueError(n)


def el(i, payload, size_len=None):
    return ebml_id(i) + ebml_size(len(payload), size_len) + payload


def uint_el(i, v):
    return el(i, v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big"))


def avcc(sps, pps, nls=4):
    return bytes([1, sps[1], sps[2], sps[3], 0xFC | (nls - 1), 0xE0 | 1]) + struct.pack(">H", len(sps)) + sps + bytes([1]) + struct.pack(">H", len(pps)) + pps


def mkv_file(codec_id, codec_private, frames, w, h, video_track=2, unknown_cluster_size=False, lacing=None):
    """frames: list of (track, bytes).  An audio track (1) precedes the video track; blocks alternate SimpleBlock / BlockGroup."""
    header = el(0x1A45DFA3, uint_el(0x4286, 1) + el(0x4282, b"matroska") + uint_el(0x4287, 4) + uint_el(0x4285, 2))
    info = el(0x1549A966, uint_el(0x2AD7B1, 1000000) + el(0x4D80, b"test") + el(0x5741, b"test"))
    audio = el(0xAE, uint_el(0xD7, 1) + uint_el(0x73C5, 11) + uint_el(0x83, 2) + el(0x86, b"A_OPUS") + el(0xE1, uint_el(0x9F, 2)))
    colour = el(0x55B0, uint_el(0x55B1, 1) + uint_el(0x55B9, 1))
    video = el(0xAE, uint_el(0xD7, video_track) + uint_el(0x73C5, 22) + uint_el(0x83, 1) + el(0x86, codec_id) + el(0x63A2, codec_private)
               + el(0xE0, uint_el(0xB0, w) + uint_el(0xBA, h) + colour))
    tracks = el(0x1654AE6B, audio + video)
    seekhead = el(0x114D9B74, el(0xEC, bytes(20)))  # void padding inside, skipped
    clusters = b""
    for ci in range(0, len(frames), 3):
        body = uint_el(0xE7, ci * 33)
        for k, (track, data) in enumerate(frames[ci:ci + 3]):
            if lacing and track == video_track and isinstance(data, list):
                n = len(data)
                assert all(len(d) == len(data[0]) for d in data)
                blk = ebml_size(track) + struct.pack(">hB", k, 0x80 | (2 << 1)) + bytes([n - 1]) + b"".join(data)
                body += el(0xA3, blk)
                continue
            blk = ebml_size(track) + struct.pack(">hB", k, 0x80) + data
            body += el(0xA3, blk) if (ci + k) % 2 == 0 else el(0xA0, el(0xA1, blk) + uint_el(0x9B, 33))
        if unknown_cluster_size:
            clusters += ebml_id(0x1F43B675) + b"\x01\xFF\xFF\xFF\xFF\xFF\xFF\xFF" + body
        else:
            clusters += el(0x1F43B675, body)
    seg_body = seekhead + info + tracks + clusters + el(0x1C53BB6B, bytes(8))
    segment = ebml_id(0x18538067) + (b"\x01\xFF\xFF\xFF\xFF\xFF\xFF\xFF" if unknown_cluster_size else ebml_size(len(seg_body), 8)) + seg_body
    return header + segment


def demux_dump(helper, path, tmp_path):
    out = str(tmp_path / "es.bin")
    head = run(helper, "demux", path, out).split()
    data = open(out, "rb").read() if os.path.exists(out) else b""
    pieces, pos = [], 0
    while pos < len(data):
        (n,) = struct.unpack_from("<I", data, pos)
        pieces.append(data[pos + 4:pos + 4 + n])
        pos += 4 + n
    return head, pieces


def test_ivf_demux(helper, tmp_path):
    seq = av1_sequence_header_obu(320, 240, color=(1, 1, 1))
    pkts = [bytes([(2 << 3) | 2, 0]) + seq + bytes([6 << 3 | 2, 3, 1, 2, 3])] + [bytes([(2 << 3) | 2, 0, 6 << 3 | 2, 2, i, i]) for i in range(4)]
    p = str(tmp_path / "a.ivf")
    open(p, "wb").write(ivf_file(b"AV01", 320, 240, pkts))
    head, pieces = demux_dump(helper, p, tmp_path)
    assert head == ["Ivf", "AV1", "5", "5"] and pieces[0] == b"" and pieces[1:] == pkts
    open(p, "wb").write(ivf_file(b"VP90", 320, 240, pkts))
    assert run(helper, "demux", p, str(tmp_path / "x")).split() == ["NONE", "IvfUnknownCodec([86,", "80,", "57,", "48])"]
    open(p, "wb").write(ivf_file(b"AV01", 320, 240, pkts)[:-3])   # truncated last packet: dropped
    assert demux_dump(helper, p, tmp_path)[0] == ["Ivf", "AV1", "5", "4"]


@pytest.mark.parametrize("unknown", [False, True])
def test_mkv_h264_demux_one_nal_unit_at_a_time(helper, tmp_path, unknown):
    sps, pps = h264_sps(640, 360, vui=(0, 1, 1, 1)), bytes([0x68, 0xCE, 0x3C, 0x80])
    nals = [[bytes([0x65, 1, 2, 3, i]), bytes([0x06, 5, i])] if i % 4 == 0 else [bytes([0x41, 9, i, i])] for i in range(7)]
    frames = []
    for i, ns in enumerate(nals):
        frames.append((1, b"opus" + bytes([i])))  # audio blocks interleaved: ignored
        frames.append((2, b"".join(struct.pack(">I", len(n)) + n for n in ns)))
    p = str(tmp_path / "a.mkv")
    open(p, "wb").write(mkv_file(b"V_MPEG4/ISO/AVC", avcc(sps, pps), frames, 640, 360, unknown_cluster_size=unknown))
    head, pieces = demux_dump(helper, p, tmp_path)
    flat = [n for ns in nals for n in ns]
    assert head == ["Mkv", "H264", "0", str(len(flat))]
    assert pieces[0] == b"\x00\x00\x00\x01" + sps + b"\x00\x00\x00\x01" + pps   # avcc_extradata_to_annexb
    assert pieces[1:] == [b"\x00\x00\x00\x01" + n for n in flat]                # avcc_into_annexb, one NAL unit per call


def test_mkv_av1_mpeg2_lacing_and_probe_errors(helper, tmp_path):
    seq = av1_sequence_header_obu(640, 360)
    av1c = bytes([0x81, 0x08, 0x0C, 0x00]) + seq
    pk = [bytes([6 << 3 | 2, 2, i, i]) for i in range(5)]
    p = str(tmp_path / "a.mkv")
    open(p, "wb").write(mkv_file(b"V_AV1", av1c, [(2, x) for x in pk], 640, 360))
    head, pieces = demux_dump(helper, p, tmp_path)
    assert head == ["Mkv", "AV1", "0", "5"] and pieces[0] == seq and pieces[1:] == pk
    m2 = mpeg2_sequence(720, 480, color=(6, 6, 6))
    open(p, "wb").write(mkv_file(b"V_MPEG2", m2, [(2, [b"\x00\x00\x01\x00ab", b"\x00\x00\x01\x00cd"]), (2, b"\x00\x00\x01\x00zz")], 720, 480, lacing="fixed"))
    head, pieces = demux_dump(helper, p, tmp_path)
    assert head == ["Mkv", "MPEG2", "0", "3"] and pieces[0] == m2 and pieces[1:] == [b"\x00\x00\x01\x00ab", b"\x00\x00\x01\x00cd", b"\x00\x00\x01\x00zz"]
    open(p, "wb").write(mkv_file(b"V_VP9", b"", [(2, b"x")], 64, 64))
    assert run(helper, "demux", p, str(tmp_path / "x")).split()[:2] == ["NONE", 'MkvUnknownCodec("V_VP9")']
    open(p, "wb").write(bytes(200))
    assert run(helper, "demux", p, str(tmp_path / "x")).split() == ["NONE", "UnknownContainer"]
    good = mkv_file(b"V_AV1", av1c, [(2, x) for x in pk], 640, 360)
    for cut in (10, 40, 90, len(good) - 7):   # truncated files end in an error or a shorter stream, never a crash
        open(p, "wb").write(good[:cut])
        r = subprocess.run([helper, "demux", p, str(tmp_path / "x")], capture_output=True, text=True)
        assert r.returncode in (0, 1) and r.stdout.split()[0] in ("NONE", "Mkv", "ERROR:"), r


FAKE_DECODER = """#!%s
import os, sys
if sys.argv[1:] == ["-version"]:
    print(os.environ.get("TM_FAKE_VERSION", "ffmpeg version 6.1.1 Copyright (c) 2000-2023 the FFmpeg developers"))
    sys.exit(0)
data = sys.stdin.buffer.read()
open(os.environ["TM_FAKE_ES"], "wb").write(" ".join(sys.argv[1:]).encode() + b"\\n" + data)
sys.stdout.buffer.write(open(os.environ["TM_FAKE_Y4M"], "rb").read())
""" % sys.executable


def test_video_source_feeds_the_decoder_and_takes_colour_from_the_sequence_header(helper, tmp_path):
    w, h, bits = 70, 38, 8
    pairs = [tm.synth.yuv420_pair(w, h, n, bits) for n in range(3)]
    y4m = str(tmp_path / "dec.y4m")
    write_y4m(y4m, [pr[0] for pr in pairs], w, h, bits)
    dec = str(tmp_path / "fakedec")
    open(dec, "w").write(FAKE_DECODER)
    os.chmod(dec, os.stat(dec).st_mode | stat.S_IXUSR)
    sps, pps = h264_sps(w, h, vui=(0, 5, 6, 5)), bytes([0x68, 0xCE, 0x3C, 0x80])
    nals = [bytes([0x65, 1, 2, 3]), bytes([0x41, 9, 9]), bytes([0x41, 7])]
    p = str(tmp_path / "v.mkv")
    open(p, "wb").write(mkv_file(b"V_MPEG4/ISO/AVC", avcc(sps, pps), [(2, struct.pack(">I", len(n)) + n) for n in nals], w, h))
    env = dict(os.environ, TM_DECODER=dec, TM_FAKE_ES=str(tmp_path / "es"), TM_FAKE_Y4M=y4m)
    out = str(tmp_path / "f.bin")
    r = subprocess.run([helper, "source", p, out], capture_output=True, text=True, env=env)
    lines = [l.split() for l in r.stdout.strip().split("\n")]
    # format id like the reference's "Mkv/H264/NVDEC"; colour from the SPS (BT.601-625 beats the 525-line fallback of a 38-row picture)
    assert lines[0] == ["Mkv/H264/fakedec", str(w), str(h), "BT601_625", "BT601_625", "BT709", "Limited", "0"], r.stdout
    assert len(lines) == 4 and lines[1][0] == "i420"
    want = b"".join(pl.astype(np.uint8).tobytes() for pr in pairs for pl in pr[0])
    assert open(out, "rb").read() == want
    es = open(env["TM_FAKE_ES"], "rb").read()
    args, stream = es.split(b"\n", 1)
    assert args == b"-nostdin -v error -f h264 -i pipe:0 -fps_mode passthrough -f yuv4mpegpipe -strict -1 pipe:1"  # one picture out per picture decoded
    assert stream == b"".join(b"\x00\x00\x00\x01" + n for n in [sps, pps] + nals)
    # ffmpeg before 5.1 (Ubuntu 22.04: 4.4, Debian 11: 4.3) has no -fps_mode: the decoder is asked for its version;
    # TM_DECODER_ARGS replaces the option for decoders that are not ffmpeg
    for version, extra_env, want_opt in (("ffmpeg version 4.4.2-0ubuntu0.22.04.1 Copyright (c) 2000-2021", {}, b"-vsync passthrough"),
                                         ("ffmpeg version 5.0.1", {}, b"-vsync passthrough"), ("ffmpeg version 5.1.4-0+deb12u1", {}, b"-fps_mode passthrough"),
                                         ("ffmpeg version n7.0.2", {}, b"-fps_mode passthrough"), ("ffmpeg version N-111111-gdeadbeef", {}, b"-fps_mode passthrough"),
                                         ("something else 1.0", {}, b"-fps_mode passthrough"), ("ffmpeg version 4.4", {"TM_DECODER_ARGS": "-an  -sn"}, b"-an -sn"),
                                         ("ffmpeg version 4.4", {"TM_DECODER_ARGS": ""}, b"")):
        r = subprocess.run([helper, "source", p, out], capture_output=True, text=True, env=dict(env, TM_FAKE_VERSION=version, **extra_env))
        assert len(r.stdout.strip().split("\n")) == 4, r.stdout
        got = open(env["TM_FAKE_ES"], "rb").read().split(b"\n", 1)[0]
        assert got == b" ".join(x for x in (b"-nostdin -v error -f h264 -i pipe:0", want_opt, b"-f yuv4mpegpipe -strict -1 pipe:1") if x), (version, got)
    # AV1 in IVF: packets re-wrapped as IVF for the decoder; size and colour from the sequence header OBU, frame count from the header
    seq = av1_sequence_header_obu(w, h, color=(1, 1, 1), full=1)
    pkts = [bytes([(2 << 3) | 2, 0]) + seq + bytes([6 << 3 | 2, 1, 7])] + [bytes([(2 << 3) | 2, 0, 6 << 3 | 2, 1, i]) for i in range(2)]
    p2 = str(tmp_path / "v.ivf")
    open(p2, "wb").write(ivf_file(b"AV01", w, h, pkts))
    r = subprocess.run([helper, "source", p2, out], capture_output=True, text=True, env=env)
    assert r.stdout.split("\n")[0].split() == ["Ivf/AV1/fakedec", str(w), str(h), "BT709", "BT709", "BT709", "Full", "3"], r.stdout
    args, stream = open(env["TM_FAKE_ES"], "rb").read().split(b"\n", 1)
    assert b"-f ivf" in args and stream[:4] == b"DKIF" and stream[32:] == b"".join(struct.pack("<IQ", len(x), i) + x for i, x in enumerate(pkts))
    # no decoder program: the error says what is missing and how to feed the CLI instead
    env["TM_DECODER"] = str(tmp_path / "no-such-decoder")
    r = subprocess.run([helper, "source", p, out], capture_output=True, text=True, env=env)
    assert r.stdout.startswith("ERROR") and "could not be started" in r.stdout and "yuv4mpegpipe" in r.stdout
    # a decoder whose output disagrees with the sequence header is refused
    write_y4m(y4m, [pr[0] for pr in [tm.synth.yuv420_pair(64, 32, 0, 8)]], 64, 32, 8)
    env["TM_DECODER"] = dec
    r = subprocess.run([helper, "source", p, out], capture_output=True, text=True, env=env)
    assert r.stdout.startswith("ERROR") and "sequence header says" in r.stdout


@pytest.mark.gpu
def test_cli_scores_a_matroska_pair_through_the_decoder_pipe(tmp_path):
    """the whole CLI on two MKV files: demuxed here, "decoded" by the stand-in decoder (started before the GPU is initialised), scored
    on the GPU -- the scores equal the Y4M run of the same pictures, the log names container / codec / decoder like the reference"""
    import json
    from tests.test_host_cli import cli
    w, h, n = 96, 64, 5
    pairs = [tm.synth.yuv420_pair(w, h, i, 8) for i in range(n)]
    dec = str(tmp_path / "fakedec")
    open(dec, "w").write(FAKE_DECODER.replace('open(os.environ["TM_FAKE_Y4M"], "rb")', 'open(os.environ["TM_FAKE_Y4M_" + ("REF" if data.find(b"REFSIDE") >= 0 else "DIS")], "rb")'))
    os.chmod(dec, os.stat(dec).st_mode | stat.S_IXUSR)
    sps, pps = h264_sps(w, h, vui=(0, 6, 6, 6)), bytes([0x68, 0xCE, 0x3C, 0x80])
    files = {}
    for side, tag in ((0, b"REFSIDE"), (1, b"DISSIDE")):
        y4m = str(tmp_path / f"{side}.y4m")
        write_y4m(y4m, [pr[side] for pr in pairs], w, h, 8)
        nals = [bytes([0x65]) + tag] + [bytes([0x41, i]) for i in range(n - 1)]
        mkv = str(tmp_path / f"{side}.mkv")
        open(mkv, "wb").write(mkv_file(b"V_MPEG4/ISO/AVC", avcc(sps, pps), [(2, struct.pack(">I", len(x)) + x) for x in nals], w, h))
        files[side] = (y4m, mkv)
    env = {"TM_DECODER": dec, "TM_FAKE_ES": str(tmp_path / "es"), "TM_FAKE_Y4M_REF": files[0][0], "TM_FAKE_Y4M_DIS": files[1][0]}
    rc, out, err = cli(files[0][1], files[1][1], "-m", "ssimulacra2", "-m", "psnr", "--output", "json", "--batch", 2, env=env)
    assert rc == 0, err
    assert "codec=Mkv/H264/fakedec" in err and "mc=BT601_525" in err and "cp=BT601_525" in err
    rc2, out2, err2 = cli(files[0][0], files[1][0], "-m", "ssimulacra2", "-m", "psnr", "--output", "json", "--batch", 2,
                          "--color-primaries", 6, "--matrix-coefficients", 6, "--transfer-characteristics", 6)
    assert rc2 == 0, err2
    assert json.loads(out) == json.loads(out2) and json.loads(out)["frame_count"] == n
    # no decoder installed: a clear failure, exit code 1
    rc, _, err = cli(files[0][1], files[1][1], "-m", "ssimulacra2", env={"TM_DECODER": str(tmp_path / "missing")})
    assert rc == 1 and "Could not read reference" in err and "could not be started" in err
