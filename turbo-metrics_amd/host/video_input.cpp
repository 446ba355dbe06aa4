// video_input.cpp -- IVF / Matroska demuxing, H.264 / AV1 / MPEG-2 sequence-header parsing and the decoder pipe.
// See video_input.hpp for what this replaces in the reference and why the decoder is an external process.
#include "video_input.hpp"

#include <cerrno>
#include <csignal>
#include <cstring>
#include <sstream>
#include <fcntl.h>
#include <stdexcept>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

namespace tm_host {

namespace {

[[noreturn]] void vfail(const std::string &m) { throw std::runtime_error(m); }

// MSB-first bit reader that never reads past the end (reads beyond it return zeros and set `over`)
struct Bits {
    const uint8_t *p;
    size_t n, pos = 0; // pos in bits
    bool over = false;
    Bits(const uint8_t *d, size_t len) : p(d), n(len) {}
    uint32_t u(int bits)
    {
        uint32_t v = 0;
        for (int i = 0; i < bits; ++i) {
            uint32_t b = 0;
            if ((pos >> 3) < n) b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
            else over = true;
            v = (v << 1) | b;
            ++pos;
        }
        return v;
    }
    uint32_t ue() // Exp-Golomb
    {
        int zeros = 0;
        while (!over && u(1) == 0 && zeros < 32) ++zeros;
        if (zeros >= 32) { over = true; return 0; }
        return zeros == 0 ? 0u : ((1u << zeros) - 1u + u(zeros));
    }
    int32_t se()
    {
        const uint32_t k = ue();
        return (k & 1u) ? (int32_t)((k + 1) / 2) : -(int32_t)(k / 2);
    }
    void skip(size_t bits) { pos += bits; if ((pos >> 3) > n) over = true; }
};

std::vector<uint8_t> unescape_rbsp(const uint8_t *p, size_t n) // drop the emulation prevention bytes (00 00 03 -> 00 00)
{
    std::vector<uint8_t> out;
    out.reserve(n);
    int zeros = 0;
    for (size_t i = 0; i < n; ++i) {
        if (zeros >= 2 && p[i] == 3) { zeros = 0; continue; }
        out.push_back(p[i]);
        zeros = p[i] == 0 ? zeros + 1 : 0;
    }
    return out;
}

const uint8_t kStartCode[4] = {0, 0, 0, 1}; // NALU_DELIMITER

} // namespace

const char *to_string(Codec c) { return c == Codec::AV1 ? "AV1" : (c == Codec::H264 ? "H264" : "MPEG2"); }
const char *to_string(Container c) { return c == Container::Mkv ? "Mkv" : "Ivf"; }

bool codec_from_fourcc(const uint8_t f[4], Codec &c)
{
    if (!memcmp(f, "AV01", 4)) { c = Codec::AV1; return true; }
    if (!memcmp(f, "AVC1", 4)) { c = Codec::H264; return true; }
    return false;
}

bool codec_from_mkv_id(const std::string &id, Codec &c)
{
    if (id == "V_MPEG4/ISO/AVC") { c = Codec::H264; return true; }
    if (id == "V_AV1") { c = Codec::AV1; return true; }
    if (id == "V_MPEG2") { c = Codec::MPEG2; return true; }
    return false;
}

// ---- H.264: sequence parameter set (ITU-T H.264 7.3.2.1.1) + VUI (E.1.1) ----------------------------------------------------
StreamFormat h264_parse_sps(const uint8_t *nal, size_t len)
{
    StreamFormat f;
    if (len < 4 || (nal[0] & 0x1F) != 7) return f;
    const std::vector<uint8_t> rbsp = unescape_rbsp(nal + 1, len - 1);
    Bits b(rbsp.data(), rbsp.size());
    const uint32_t profile = b.u(8);
    b.u(8); // constraint_set flags + reserved
    b.u(8); // level_idc
    b.ue(); // seq_parameter_set_id
    uint32_t chroma = 1, depth = 8;
    bool separate = false;
    static const uint32_t high[] = {100, 110, 122, 244, 44, 83, 86, 118, 128, 138, 139, 134, 135};
    bool is_high = false;
    for (uint32_t h : high) is_high |= h == profile;
    if (is_high) {
        chroma = b.ue();
        if (chroma == 3) separate = b.u(1) != 0;
        depth = 8 + b.ue();
        b.ue();  // bit_depth_chroma_minus8
        b.u(1);  // qpprime_y_zero_transform_bypass_flag
        if (b.u(1)) { // seq_scaling_matrix_present_flag
            const int lists = chroma != 3 ? 8 : 12;
            for (int i = 0; i < lists; ++i)
                if (b.u(1)) {
                    int last = 8, next = 8;
                    const int size = i < 6 ? 16 : 64;
                    for (int j = 0; j < size; ++j) {
                        if (next != 0) next = (last + b.se() + 256) % 256;
                        last = next == 0 ? last : next;
                    }
                }
        }
    }
    b.ue(); // log2_max_frame_num_minus4
    const uint32_t poc_type = b.ue();
    if (poc_type == 0) b.ue();
    else if (poc_type == 1) {
        b.u(1); b.se(); b.se();
        const uint32_t cyc = b.ue();
        if (cyc > 255) return f;
        for (uint32_t i = 0; i < cyc; ++i) b.se();
    }
    b.ue(); // max_num_ref_frames
    b.u(1); // gaps_in_frame_num_value_allowed_flag
    const uint32_t mbs_w = b.ue() + 1, map_h = b.ue() + 1;
    const uint32_t frame_mbs_only = b.u(1);
    if (!frame_mbs_only) b.u(1); // mb_adaptive_frame_field_flag
    b.u(1); // direct_8x8_inference_flag
    uint32_t crop[4] = {0, 0, 0, 0}; // left, right, top, bottom
    if (b.u(1))
        for (uint32_t &c : crop) c = b.ue();
    if (b.over || chroma > 3 || depth > 14 || mbs_w > 4096 || map_h > 4096) return f;
    const uint32_t chroma_array = separate ? 0 : chroma;
    const uint32_t unit_x = chroma_array == 0 ? 1 : (chroma_array == 3 ? 1 : 2);
    const uint32_t unit_y = (chroma_array == 1 ? 2 : 1) * (2 - frame_mbs_only);
    const uint64_t cw = (uint64_t)unit_x * (crop[0] + crop[1]), chh = (uint64_t)unit_y * (crop[2] + crop[3]);
    const uint64_t w = (uint64_t)mbs_w * 16, h = (uint64_t)(2 - frame_mbs_only) * map_h * 16;
    if (cw >= w || chh >= h) return f;
    f.width = (uint32_t)(w - cw); f.height = (uint32_t)(h - chh);
    f.bit_depth = (int)depth; f.chroma_format = (int)chroma;
    if (b.u(1)) { // vui_parameters_present_flag
        if (b.u(1)) { // aspect_ratio_info_present_flag
            if (b.u(8) == 255) b.skip(32);
        }
        if (b.u(1)) b.u(1); // overscan
        if (b.u(1)) {       // video_signal_type_present_flag
            b.u(3);         // video_format
            f.full_range = b.u(1) != 0;
            if (b.u(1)) { f.cp = (int)b.u(8); f.tc = (int)b.u(8); f.mc = (int)b.u(8); }
        }
    }
    f.valid = !b.over;
    return f;
}

// ---- AV1: sequence header OBU (AV1 bitstream spec 5.5) ------------------------------------------------------------------------
namespace {

bool leb128(const uint8_t *p, size_t n, size_t &pos, uint64_t &v)
{
    v = 0;
    for (int i = 0; i < 8; ++i) {
        if (pos >= n) return false;
        const uint8_t b = p[pos++];
        v |= (uint64_t)(b & 0x7F) << (7 * i);
        if (!(b & 0x80)) return true;
    }
    return false;
}

uint32_t uvlc(Bits &b)
{
    int zeros = 0;
    while (!b.over && b.u(1) == 0 && zeros < 32) ++zeros;
    if (zeros >= 32) return 0xFFFFFFFFu;
    return zeros == 0 ? 0u : (b.u(zeros) + (1u << zeros) - 1u);
}

StreamFormat av1_sequence_header_payload(const uint8_t *p, size_t n)
{
    StreamFormat f;
    Bits b(p, n);
    const uint32_t profile = b.u(3);
    b.u(1); // still_picture
    const uint32_t reduced = b.u(1);
    if (reduced) {
        b.u(5); // seq_level_idx[0]
    } else {
        uint32_t decoder_model = 0, buffer_delay_len = 0;
        if (b.u(1)) { // timing_info_present_flag
            b.skip(64); // num_units_in_display_tick, time_scale
            if (b.u(1)) uvlc(b); // equal_picture_interval -> num_ticks_per_picture_minus_1
            decoder_model = b.u(1);
            if (decoder_model) {
                buffer_delay_len = b.u(5) + 1;
                b.skip(32); // num_units_in_decoding_tick
                b.u(5); b.u(5); // buffer_removal_time_length_minus_1, frame_presentation_time_length_minus_1
            }
        }
        const uint32_t initial_display_delay = b.u(1);
        const uint32_t ops = b.u(5) + 1;
        for (uint32_t i = 0; i < ops; ++i) {
            b.u(12); // operating_point_idc
            if (b.u(5) > 7) b.u(1); // seq_level_idx, seq_tier
            if (decoder_model && b.u(1)) { b.skip(2 * (size_t)buffer_delay_len); b.u(1); }
            if (initial_display_delay && b.u(1)) b.u(4);
        }
    }
    const uint32_t wbits = b.u(4) + 1, hbits = b.u(4) + 1;
    f.width = b.u((int)wbits) + 1;
    f.height = b.u((int)hbits) + 1;
    if (!reduced) {
        if (b.u(1)) { b.u(4); b.u(3); } // frame_id_numbers_present_flag
    }
    b.u(1); b.u(1); b.u(1); // use_128x128_superblock, enable_filter_intra, enable_intra_edge_filter
    if (!reduced) {
        b.u(4); // enable_interintra_compound, enable_masked_compound, enable_warped_motion, enable_dual_filter
        const uint32_t order_hint = b.u(1);
        if (order_hint) b.u(2); // enable_jnt_comp, enable_ref_frame_mvs
        uint32_t force_sct = 2;  // SELECT_SCREEN_CONTENT_TOOLS
        if (!b.u(1)) force_sct = b.u(1); // seq_choose_screen_content_tools
        if (force_sct > 0) { if (!b.u(1)) b.u(1); } // seq_choose_integer_mv / seq_force_integer_mv
        if (order_hint) b.u(3);
    }
    b.u(3); // enable_superres, enable_cdef, enable_restoration
    // color_config()
    const uint32_t high = b.u(1);
    uint32_t depth = high ? 10 : 8;
    if (profile == 2 && high && b.u(1)) depth = 12;
    const uint32_t mono = profile == 1 ? 0 : b.u(1);
    if (b.u(1)) { f.cp = (int)b.u(8); f.tc = (int)b.u(8); f.mc = (int)b.u(8); }
    uint32_t ssx = 1, ssy = 1;
    if (mono) {
        f.full_range = b.u(1) != 0;
    } else if (f.cp == 1 && f.tc == 13 && f.mc == 0) { // sRGB / identity: 4:4:4, full range
        f.full_range = true; ssx = ssy = 0;
    } else {
        f.full_range = b.u(1) != 0;
        if (profile == 0) { ssx = ssy = 1; }
        else if (profile == 1) { ssx = ssy = 0; }
        else if (depth == 12) { ssx = b.u(1); ssy = ssx ? b.u(1) : 0; }
        else { ssx = 1; ssy = 0; }
    }
    f.bit_depth = (int)depth;
    f.chroma_format = mono ? 0 : (ssx && ssy ? 1 : (ssx ? 2 : 3));
    f.valid = !b.over && profile <= 2;
    return f;
}

} // namespace

StreamFormat av1_parse_sequence_header(const uint8_t *data, size_t len)
{
    size_t pos = 0;
    while (pos < len) {
        const uint8_t hdr = data[pos++];
        if (hdr & 0x80) break; // forbidden bit
        const int type = (hdr >> 3) & 0xF;
        if (hdr & 0x04) { if (pos >= len) break; ++pos; } // extension byte
        uint64_t size = len - pos;
        if (hdr & 0x02) { if (!leb128(data, len, pos, size)) break; }
        if (size > len - pos) break;
        if (type == 1) return av1_sequence_header_payload(data + pos, (size_t)size);
        pos += (size_t)size;
    }
    return StreamFormat{};
}

// ---- MPEG-2 video: sequence_header, sequence_extension, sequence_display_extension (ISO/IEC 13818-2 6.2.2) -------------------
StreamFormat mpeg2_parse_sequence(const uint8_t *d, size_t n)
{
    StreamFormat f;
    auto find = [&](size_t from, uint8_t code) -> size_t {
        for (size_t i = from; i + 3 < n; ++i)
            if (d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 1 && d[i + 3] == code) return i + 4;
        return (size_t)-1;
    };
    size_t at = find(0, 0xB3);
    if (at == (size_t)-1) return f;
    Bits b(d + at, n - at);
    uint32_t w = b.u(12), h = b.u(12);
    b.u(4); b.u(4); b.u(18); b.u(1); b.u(10); b.u(1);
    if (b.u(1)) b.skip(512);
    if (b.u(1)) b.skip(512);
    if (b.over || w == 0 || h == 0) return f;
    f.chroma_format = 1;
    for (size_t e = find(at, 0xB5); e != (size_t)-1; e = find(e, 0xB5)) {
        Bits x(d + e, n - e);
        const uint32_t id = x.u(4);
        if (id == 1) { // sequence_extension
            x.u(8); x.u(1);
            f.chroma_format = (int)x.u(2);
            w |= x.u(2) << 12; h |= x.u(2) << 12;
        } else if (id == 2) { // sequence_display_extension
            x.u(3);
            if (x.u(1)) { f.cp = (int)x.u(8); f.tc = (int)x.u(8); f.mc = (int)x.u(8); }
        }
        if (x.over) break;
    }
    f.width = w; f.height = h; f.bit_depth = 8;
    f.valid = true;
    return f;
}

// ---- avcC -------------------------------------------------------------------------------------------------------------------
size_t avcc_extradata_to_annexb(const std::vector<uint8_t> &cp, std::vector<uint8_t> &out)
{
    out.clear();
    if (cp.size() < 7) return 0;
    const size_t nal_size = (size_t)(cp[4] & 3) + 1;
    size_t pos = 5;
    auto sets = [&](size_t count) {
        for (size_t i = 0; i < count; ++i) {
            if (pos + 2 > cp.size()) return false;
            const size_t len = (size_t)cp[pos] << 8 | cp[pos + 1];
            pos += 2;
            if (pos + len > cp.size()) return false;
            out.insert(out.end(), kStartCode, kStartCode + 4);
            out.insert(out.end(), cp.begin() + (long)pos, cp.begin() + (long)(pos + len));
            pos += len;
        }
        return true;
    };
    if (!sets(cp[pos++] & 0x1F)) return 0;
    if (pos >= cp.size()) return 0;
    if (!sets(cp[pos++])) return 0;
    return nal_size;
}

size_t avcc_into_annexb(const uint8_t *buf, size_t len, size_t nls, std::vector<uint8_t> &nalu)
{
    if (nls == 0 || nls > 4 || len <= nls) return 0;
    size_t n = 0;
    for (size_t i = 0; i < nls; ++i) n = (n << 8) | buf[i];
    if (n > len - nls) return 0;
    nalu.assign(kStartCode, kStartCode + 4);
    nalu.insert(nalu.end(), buf + nls, buf + nls + n);
    return nls + n;
}

// ---- IVF --------------------------------------------------------------------------------------------------------------------
bool ivf_read_header(const uint8_t *p, size_t n, IvfHeader &h, size_t &header_len)
{
    if (n < 32 || memcmp(p, "DKIF", 4)) return false;
    auto le16 = [&](size_t o) { return (uint16_t)(p[o] | p[o + 1] << 8); };
    auto le32 = [&](size_t o) { return (uint32_t)p[o] | (uint32_t)p[o + 1] << 8 | (uint32_t)p[o + 2] << 16 | (uint32_t)p[o + 3] << 24; };
    header_len = le16(6);
    memcpy(h.fourcc, p + 8, 4);
    h.w = le16(12); h.h = le16(14);
    h.timebase_den = le32(16); h.timebase_num = le32(20);
    h.frames = le32(24);
    return header_len >= 32;
}

namespace {

class IvfDemuxer : public Demuxer {
public:
    IvfDemuxer(FILE *f, const IvfHeader &h, Codec c) : f_(f), h_(h), codec_(c) {}
    ~IvfDemuxer() override { if (f_ && f_ != stdin) fclose(f_); }
    Container container() const override { return Container::Ivf; }
    Codec codec() const override { return codec_; }
    size_t frame_count() const override { return h_.frames; }
    void init(std::vector<uint8_t> &out) override { out.clear(); } // IVF carries no out-of-band data: the first packet starts the stream
    bool demux(std::vector<uint8_t> &out) override
    {
        uint8_t fh[12];
        if (fread(fh, 1, 12, f_) != 12) return false;
        const uint32_t len = (uint32_t)fh[0] | (uint32_t)fh[1] << 8 | (uint32_t)fh[2] << 16 | (uint32_t)fh[3] << 24;
        if (len == 0 || len > (256u << 20)) return false;
        out.resize(len);
        return fread(out.data(), 1, len, f_) == len;
    }

private:
    FILE *f_;
    IvfHeader h_;
    Codec codec_;
};

// ---- Matroska (EBML) ---------------------------------------------------------------------------------------------------------
constexpr uint64_t kUnknownSize = ~0ull;
enum : uint32_t {
    ID_EBML = 0x1A45DFA3, ID_SEGMENT = 0x18538067, ID_INFO = 0x1549A966, ID_TRACKS = 0x1654AE6B, ID_CLUSTER = 0x1F43B675,
    ID_TRACK_ENTRY = 0xAE, ID_TRACK_NUMBER = 0xD7, ID_TRACK_TYPE = 0x83, ID_CODEC_ID = 0x86, ID_CODEC_PRIVATE = 0x63A2, ID_VIDEO = 0xE0,
    ID_SIMPLE_BLOCK = 0xA3, ID_BLOCK_GROUP = 0xA0, ID_BLOCK = 0xA1, ID_CUES = 0x1C53BB6B, ID_TAGS = 0x1254C367, ID_SEEKHEAD = 0x114D9B74,
    ID_ATTACHMENTS = 0x1941A469, ID_CHAPTERS = 0x1043A770,
};

struct MkvTrack {
    uint64_t number = 0, type = 0;
    std::string codec_id;
    std::vector<uint8_t> codec_private;
    bool has_video = false;
};

class MkvDemuxer : public Demuxer {
public:
    explicit MkvDemuxer(FILE *f) : f_(f) {}
    ~MkvDemuxer() override { if (f_ && f_ != stdin) fclose(f_); }
    // parses up to the first Cluster; "" on success, else the reference's ProbeError text
    std::string open()
    {
        uint32_t id; uint64_t size;
        if (!element(id, size) || id != ID_EBML || size == kUnknownSize || size > (1u << 20)) return "UnknownContainer";
        if (!skip(size)) return "UnknownContainer";
        if (!element(id, size) || id != ID_SEGMENT) return "UnknownContainer";
        segment_end_ = size == kUnknownSize ? kUnknownSize : (uint64_t)ftello(f_) + size;
        std::vector<MkvTrack> tracks;
        for (;;) {
            const off_t at = ftello(f_);
            if (segment_end_ != kUnknownSize && (uint64_t)at >= segment_end_) break;
            if (!element(id, size)) break;
            if (id == ID_CLUSTER) { first_cluster_ = at; break; }
            if (id == ID_TRACKS) {
                if (size == kUnknownSize || size > (64u << 20)) return "UnknownContainer";
                std::vector<uint8_t> buf((size_t)size);
                if (fread(buf.data(), 1, buf.size(), f_) != buf.size()) return "UnknownContainer";
                parse_tracks(buf.data(), buf.size(), tracks);
            } else {
                if (size == kUnknownSize || !skip(size)) return "UnknownContainer";
            }
        }
        const MkvTrack *video = nullptr;
        for (const MkvTrack &t : tracks)
            if (t.has_video) { video = &t; break; } // mkv_find_video_track: the first track with a Video element
        if (!video) return "MKVNoVideo";
        if (!codec_from_mkv_id(video->codec_id, codec_)) return "MkvUnknownCodec(\"" + video->codec_id + "\")";
        track_ = *video;
        if (first_cluster_ < 0) return "UnknownContainer";
        fseeko(f_, first_cluster_, SEEK_SET);
        return "";
    }
    void release() { f_ = nullptr; }
    Container container() const override { return Container::Mkv; }
    Codec codec() const override { return codec_; }
    size_t frame_count() const override { return 0; }
    void init(std::vector<uint8_t> &out) override
    {
        out.clear();
        switch (codec_) {
        case Codec::MPEG2: out = track_.codec_private; break; // input_video.rs:273-278
        case Codec::H264:
            nal_length_size_ = avcc_extradata_to_annexb(track_.codec_private, out);
            if (nal_length_size_ == 0) vfail("MKV: malformed avcC CodecPrivate");
            break;
        case Codec::AV1: // av1::extract_seq_hdr_from_mkv_codec_private: past the 4 bytes of the av1C box
            if (track_.codec_private.size() > 4) out.assign(track_.codec_private.begin() + 4, track_.codec_private.end());
            break;
        }
    }
    bool demux(std::vector<uint8_t> &out) override
    {
        for (;;) {
            if (frame_off_ < frame_.size()) {
                if (codec_ != Codec::H264) { out = frame_; frame_off_ = frame_.size(); return true; }
                // H.264: one NAL unit at a time (input_video.rs:311-331)
                const size_t used = avcc_into_annexb(frame_.data() + frame_off_, frame_.size() - frame_off_, nal_length_size_, out);
                if (used == 0) vfail("incomplete nalu");
                frame_off_ += used;
                return true;
            }
            if (!laced_.empty()) { frame_ = std::move(laced_.front()); laced_.erase(laced_.begin()); frame_off_ = 0; continue; }
            if (!next_block()) return false;
        }
    }

private:
    bool vint(uint64_t &v, bool keep_marker, int max_len, int *len_out = nullptr)
    {
        const int c = fgetc(f_);
        if (c == EOF) return false;
        int len = 1;
        while (len <= max_len && !(c & (0x80 >> (len - 1)))) ++len;
        if (len > max_len) return false;
        v = keep_marker ? (uint64_t)c : (uint64_t)(c & (0xFF >> len));
        bool all_ones = (c & (0xFF >> len)) == (0xFF >> len);
        for (int i = 1; i < len; ++i) {
            const int b = fgetc(f_);
            if (b == EOF) return false;
            v = (v << 8) | (uint64_t)b;
            all_ones &= b == 0xFF;
        }
        if (!keep_marker && all_ones) v = kUnknownSize;
        if (len_out) *len_out = len;
        return true;
    }
    bool element(uint32_t &id, uint64_t &size)
    {
        uint64_t v;
        if (!vint(v, true, 4)) return false;
        id = (uint32_t)v;
        return vint(size, false, 8);
    }
    bool skip(uint64_t n) { return fseeko(f_, (off_t)n, SEEK_CUR) == 0; }

    // in-memory EBML walk (Tracks is read whole)
    static bool mem_vint(const uint8_t *p, size_t n, size_t &pos, uint64_t &v, bool keep_marker, int max_len)
    {
        if (pos >= n) return false;
        const uint8_t c = p[pos];
        int len = 1;
        while (len <= max_len && !(c & (0x80 >> (len - 1)))) ++len;
        if (len > max_len || pos + (size_t)len > n) return false;
        v = keep_marker ? c : (uint64_t)(c & (0xFF >> len));
        for (int i = 1; i < len; ++i) v = (v << 8) | p[pos + (size_t)i];
        pos += (size_t)len;
        return true;
    }
    static uint64_t be_uint(const uint8_t *p, size_t n) { uint64_t v = 0; for (size_t i = 0; i < n && i < 8; ++i) v = (v << 8) | p[i]; return v; }
    static void parse_tracks(const uint8_t *p, size_t n, std::vector<MkvTrack> &out)
    {
        size_t pos = 0;
        while (pos < n) {
            uint64_t id, size;
            if (!mem_vint(p, n, pos, id, true, 4) || !mem_vint(p, n, pos, size, false, 8) || size > n - pos) return;
            if (id == ID_TRACK_ENTRY) {
                MkvTrack t;
                size_t q = pos;
                const size_t end = pos + (size_t)size;
                while (q < end) {
                    uint64_t cid, cs;
                    if (!mem_vint(p, end, q, cid, true, 4) || !mem_vint(p, end, q, cs, false, 8) || cs > end - q) break;
                    if (cid == ID_TRACK_NUMBER) t.number = be_uint(p + q, (size_t)cs);
                    else if (cid == ID_TRACK_TYPE) t.type = be_uint(p + q, (size_t)cs);
                    else if (cid == ID_CODEC_ID) t.codec_id.assign((const char *)p + q, (size_t)cs);
                    else if (cid == ID_CODEC_PRIVATE) t.codec_private.assign(p + q, p + q + cs);
                    else if (cid == ID_VIDEO) t.has_video = true;
                    q += (size_t)cs;
                }
                while (!t.codec_id.empty() && t.codec_id.back() == '\0') t.codec_id.pop_back();
                out.push_back(std::move(t));
            }
            pos += (size_t)size;
        }
    }

    // the next Block / SimpleBlock of the video track -> frame_ (+ laced_)
    bool next_block()
    {
        for (;;) {
            const off_t at = ftello(f_);
            if (segment_end_ != kUnknownSize && (uint64_t)at >= segment_end_) return false;
            uint32_t id; uint64_t size;
            if (!element(id, size)) return false;
            if (id == ID_CLUSTER || id == ID_BLOCK_GROUP) continue; // descend: their children follow
            if (id == ID_SIMPLE_BLOCK || id == ID_BLOCK) {
                if (size == kUnknownSize || size > (512u << 20) || size < 4) return false;
                std::vector<uint8_t> buf((size_t)size);
                if (fread(buf.data(), 1, buf.size(), f_) != buf.size()) return false;
                size_t pos = 0;
                uint64_t track;
                if (!mem_vint(buf.data(), buf.size(), pos, track, false, 8) || pos + 3 > buf.size()) return false;
                const uint8_t flags = buf[pos + 2];
                pos += 3;
                if (track != track_.number) continue;
                const int lacing = (flags >> 1) & 3;
                if (lacing == 0) { frame_.assign(buf.begin() + (long)pos, buf.end()); frame_off_ = 0; return true; }
                if (!unlace(buf, pos, lacing)) return false;
                if (laced_.empty()) continue;
                frame_ = std::move(laced_.front()); laced_.erase(laced_.begin()); frame_off_ = 0;
                return true;
            }
            if (size == kUnknownSize) return false; // only Segment / Cluster may be of unknown size
            if (!skip(size)) return false;
        }
    }
    bool unlace(const std::vector<uint8_t> &buf, size_t pos, int lacing)
    {
        if (pos >= buf.size()) return false;
        const size_t count = (size_t)buf[pos++] + 1;
        std::vector<size_t> sizes(count, 0);
        if (lacing == 2) { // fixed size
            const size_t total = buf.size() - pos;
            if (total % count) return false;
            for (size_t &s : sizes) s = total / count;
        } else if (lacing == 1) { // Xiph
            size_t sum = 0;
            for (size_t i = 0; i + 1 < count; ++i) {
                size_t s = 0;
                for (;;) { if (pos >= buf.size()) return false; const uint8_t b = buf[pos++]; s += b; if (b != 255) break; }
                sizes[i] = s; sum += s;
            }
            if (sum > buf.size() - pos) return false;
            sizes[count - 1] = buf.size() - pos - sum;
        } else { // EBML
            uint64_t v; size_t sum = 0;
            if (!mem_vint(buf.data(), buf.size(), pos, v, false, 8)) return false;
            sizes[0] = (size_t)v; sum = sizes[0];
            for (size_t i = 1; i + 1 < count; ++i) {
                const size_t before = pos;
                if (!mem_vint(buf.data(), buf.size(), pos, v, false, 8)) return false;
                const int len = (int)(pos - before);
                const int64_t delta = (int64_t)v - (((int64_t)1 << (7 * len - 1)) - 1);
                const int64_t s = (int64_t)sizes[i - 1] + delta;
                if (s < 0) return false;
                sizes[i] = (size_t)s; sum += sizes[i];
            }
            if (sum > buf.size() - pos) return false;
            if (count > 1) sizes[count - 1] = buf.size() - pos - sum;
        }
        for (size_t s : sizes) {
            if (s > buf.size() - pos) return false;
            laced_.emplace_back(buf.begin() + (long)pos, buf.begin() + (long)(pos + s));
            pos += s;
        }
        return true;
    }

    FILE *f_;
    uint64_t segment_end_ = kUnknownSize;
    off_t first_cluster_ = -1;
    MkvTrack track_;
    Codec codec_ = Codec::H264;
    size_t nal_length_size_ = 0;
    std::vector<uint8_t> frame_;
    size_t frame_off_ = 0;
    std::vector<std::vector<uint8_t>> laced_;
};

} // namespace

std::unique_ptr<Demuxer> probe_video(FILE *f, std::string &why)
{
    uint8_t head[64];
    const long at = ftell(f);
    const size_t got = fread(head, 1, sizeof head, f);
    IvfHeader ih;
    size_t hl = 0;
    if (ivf_read_header(head, got, ih, hl)) {
        Codec c;
        if (!codec_from_fourcc(ih.fourcc, c)) {
            char b[64];
            snprintf(b, sizeof b, "IvfUnknownCodec([%u, %u, %u, %u])", ih.fourcc[0], ih.fourcc[1], ih.fourcc[2], ih.fourcc[3]);
            why = b;
            return nullptr;
        }
        // position the stream right after the header (a pipe cannot seek: the header is 32 bytes and 64 were read -> only files)
        if (at < 0 || fseek(f, at + (long)hl, SEEK_SET) != 0) { why = "UnknownContainer"; return nullptr; }
        return std::make_unique<IvfDemuxer>(f, ih, c);
    }
    if (at < 0 || fseek(f, at, SEEK_SET) != 0) { why = "UnknownContainer"; return nullptr; } // Matroska needs a seekable file (like the reference)
    auto mkv = std::make_unique<MkvDemuxer>(f);
    why = mkv->open();
    if (!why.empty()) {
        mkv->release(); // the caller keeps the stream on failure
        return nullptr;
    }
    return mkv;
}

// ---- the decoder pipe --------------------------------------------------------------------------------------------------------
namespace {

void write_all(int fd, const uint8_t *p, size_t n)
{
    while (n > 0) {
        const ssize_t w = write(fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; return; } // EPIPE: the decoder went away; the reader will notice
        p += w; n -= (size_t)w;
    }
}

void put_le(std::vector<uint8_t> &v, uint64_t x, int bytes) { for (int i = 0; i < bytes; ++i) v.push_back((uint8_t)(x >> (8 * i))); }

} // namespace

// the decoder's option for "no frame-rate conversion": see the call site
std::vector<std::string> decoder_sync_args(const std::string &prog)
{
    std::vector<std::string> out;
    if (const char *env = getenv("TM_DECODER_ARGS")) {
        std::istringstream ss(env);
        for (std::string t; ss >> t;) out.push_back(t);
        return out;
    }
    int major = -1, minor = 0;
    bool safe = !prog.empty();
    for (char c : prog) safe = safe && (isalnum((unsigned char)c) || strchr("/._-+", c)); // the name goes through a shell below
    if (safe) {
        if (FILE *p = popen((prog + " -version 2>/dev/null </dev/null").c_str(), "r")) {
            char line[256];
            if (fgets(line, sizeof line, p)) { // "ffmpeg version 4.4.2-0ubuntu0.22.04.1 Copyright ..." / "ffmpeg version n6.1.1" / "ffmpeg version N-110000-g..."
                const char *v = strstr(line, "version ");
                if (v) {
                    v += 8;
                    if (*v == 'n') ++v;
                    if (isdigit((unsigned char)*v)) { major = atoi(v); const char *dot = strchr(v, '.'); minor = dot ? atoi(dot + 1) : 0; }
                }
            }
            pclose(p);
        }
    }
    if (major >= 0 && (major < 5 || (major == 5 && minor < 1))) out = {"-vsync", "passthrough"};
    else out = {"-fps_mode", "passthrough"}; // 5.1 and later, git snapshots, and anything that does not say
    return out;
}

VideoFrameSource::VideoFrameSource(std::unique_ptr<Demuxer> demuxer, const SourceHints &hints) : demuxer_(std::move(demuxer))
{
    // sequence header: out-of-band data first, then packets until one carries it (what VideoFrameSource::new does with cuvid's
    // format callback, input_video.rs:362-368); everything pulled here is fed to the decoder later
    std::vector<std::vector<uint8_t>> pending;
    std::vector<uint8_t> extra;
    demuxer_->init(extra);
    const Codec codec = demuxer_->codec();
    auto try_format = [&](const std::vector<uint8_t> &d) {
        if (fmt_.valid || d.empty()) return;
        if (codec == Codec::AV1) fmt_ = av1_parse_sequence_header(d.data(), d.size());
        else if (codec == Codec::MPEG2) fmt_ = mpeg2_parse_sequence(d.data(), d.size());
        else { // Annex B: every NAL unit after a start code
            for (size_t i = 0; i + 4 < d.size() && !fmt_.valid; ++i)
                if (d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 1 && (d[i + 3] & 0x1F) == 7) {
                    size_t end = d.size();
                    for (size_t j = i + 3; j + 2 < d.size(); ++j)
                        if (d[j] == 0 && d[j + 1] == 0 && (d[j + 2] == 1 || (d[j + 2] == 0 && j + 3 < d.size() && d[j + 3] == 1))) { end = j; break; }
                    fmt_ = h264_parse_sps(d.data() + i + 3, end - (i + 3));
                }
        }
    };
    try_format(extra);
    for (int i = 0; i < 64 && !fmt_.valid; ++i) {
        std::vector<uint8_t> pkt;
        if (!demuxer_->demux(pkt)) break;
        try_format(pkt);
        pending.push_back(std::move(pkt));
    }
    if (!fmt_.valid) vfail(std::string("no ") + to_string(codec) + " sequence header found in the first packets");
    if (fmt_.chroma_format != 1)
        vfail("not implemented: only 4:2:0 video reaches the NV12 / P016 surfaces of the reference (chroma_format_idc " + std::to_string(fmt_.chroma_format) + ")");

    // ---- the decoder process: elementary stream on its stdin, YUV4MPEG2 on its stdout
    const char *prog = getenv("TM_DECODER");
    decoder_name_ = prog && *prog ? prog : "ffmpeg";
    const char *in_fmt = codec == Codec::H264 ? "h264" : (codec == Codec::MPEG2 ? "mpegvideo" : "ivf"); // AV1 packets are re-wrapped as IVF
    // "one picture out per picture decoded" is `-fps_mode passthrough` from ffmpeg 5.1 on and `-vsync passthrough` before (Ubuntu 22.04
    // ships 4.4, Debian 11 4.3; 7.0 removed -vsync): the decoder is asked for its version first.  TM_DECODER_ARGS replaces
    // the option altogether (whitespace-separated; may be empty) for decoders that are not ffmpeg.
    const std::vector<std::string> sync_args = decoder_sync_args(decoder_name_);
    int to_child[2], from_child[2];
    if (pipe2(to_child, O_CLOEXEC) != 0 || pipe2(from_child, O_CLOEXEC) != 0) vfail(std::string("pipe: ") + strerror(errno));
    signal(SIGPIPE, SIG_IGN);
    const pid_t pid = fork();
    if (pid < 0) vfail(std::string("fork: ") + strerror(errno));
    if (pid == 0) {
        dup2(to_child[0], 0); dup2(from_child[1], 1);
        close(to_child[0]); close(to_child[1]); close(from_child[0]); close(from_child[1]);
        // one output picture per decoded picture, whatever rate the muxer guesses (raw streams carry no timestamps): without
        // passthrough a constant-frame-rate muxer may duplicate or drop pictures and misalign the reference / distorted pairs
        // (the reference's NVDEC path emits exactly one picture per decode).  dup2 clears O_CLOEXEC on descriptors 0 and 1.
        std::vector<const char *> argv = {decoder_name_.c_str(), "-nostdin", "-v", "error", "-f", in_fmt, "-i", "pipe:0"};
        for (const std::string &a : sync_args) argv.push_back(a.c_str());
        for (const char *a : {"-f", "yuv4mpegpipe", "-strict", "-1", "pipe:1"}) argv.push_back(a);
        argv.push_back(nullptr);
        execvp(decoder_name_.c_str(), (char *const *)argv.data());
        _exit(127);
    }
    child_ = (int)pid;
    close(to_child[0]); close(from_child[1]);
    const int wfd = to_child[1];
    Demuxer *dm = demuxer_.get();
    const uint32_t w = fmt_.width, h = fmt_.height;
    feeder_ = std::thread([dm, wfd, codec, extra = std::move(extra), pending = std::move(pending), w, h]() mutable {
        uint64_t n = 0;
        std::vector<uint8_t> hdr;
        auto send = [&](const std::vector<uint8_t> &pkt) {
            if (pkt.empty()) return;
            if (codec == Codec::AV1) { hdr.clear(); put_le(hdr, pkt.size(), 4); put_le(hdr, n, 8); write_all(wfd, hdr.data(), hdr.size()); }
            write_all(wfd, pkt.data(), pkt.size());
            ++n;
        };
        if (codec == Codec::AV1) { // IVF file header
            std::vector<uint8_t> fh = {'D', 'K', 'I', 'F'};
            put_le(fh, 0, 2); put_le(fh, 32, 2);
            fh.insert(fh.end(), {'A', 'V', '0', '1'});
            put_le(fh, w, 2); put_le(fh, h, 2); put_le(fh, 1000000, 4); put_le(fh, 1, 4); put_le(fh, 0, 4); put_le(fh, 0, 4);
            write_all(wfd, fh.data(), fh.size());
            // the out-of-band sequence header is repeated in-band by every AV1 key frame: not sent on its own
        } else send(extra);
        for (const auto &p : pending) send(p);
        std::vector<uint8_t> pkt;
        try {
            while (dm->demux(pkt)) send(pkt);
        } catch (...) {} // a malformed tail ends the stream; the reader sees the end of the decoder's output
        close(wfd);
    });
    FILE *out = fdopen(from_child[0], "rb");
    if (!out) { close(from_child[0]); shutdown(); vfail("fdopen failed"); }
    try {
        inner_ = open_y4m_stream(out, "", false, hints, "the decoder's output");
    } catch (const std::exception &e) {
        int status = 0;
        const bool gone = waitpid((pid_t)child_, &status, WNOHANG) == (pid_t)child_;
        const bool missing = gone && WIFEXITED(status) && WEXITSTATUS(status) == 127;
        if (gone) child_ = -1;
        fclose(out);
        if (feeder_.joinable()) feeder_.join();
        if (child_ > 0) { kill((pid_t)child_, SIGTERM); waitpid((pid_t)child_, nullptr, 0); child_ = -1; }
        if (missing)
            vfail(std::string(to_string(demuxer_->container())) + "/" + to_string(codec) + " input needs a decoder and `" + decoder_name_ +
                  "` could not be started: MI355X has no video decode engine behind HIP and none is built in; install ffmpeg (or point TM_DECODER at a "
                  "program with its command line), or feed YUV4MPEG2 (`ffmpeg -i in.mkv -f yuv4mpegpipe - | turbo-metrics - ...`)");
        vfail(std::string("decoder `") + decoder_name_ + "` produced no YUV4MPEG2 stream: " + e.what());
    }
    if (inner_->width() != fmt_.width || inner_->height() != fmt_.height) {
        const std::string msg = "decoder output is " + std::to_string(inner_->width()) + "x" + std::to_string(inner_->height()) +
                                ", the sequence header says " + std::to_string(fmt_.width) + "x" + std::to_string(fmt_.height);
        shutdown(); // a constructor that throws gets no destructor call: the feeder thread and the child are taken down here
        vfail(msg);
    }
}

void VideoFrameSource::shutdown()
{
    inner_.reset(); // closes the read end: a decoder still writing gets EPIPE
    if (child_ > 0) {
        int status = 0;
        if (waitpid((pid_t)child_, &status, WNOHANG) == 0) { kill((pid_t)child_, SIGTERM); waitpid((pid_t)child_, &status, 0); }
        child_ = -1;
    }
    if (feeder_.joinable()) feeder_.join(); // its write end fails with EPIPE once the child is gone
}

VideoFrameSource::~VideoFrameSource() { shutdown(); }

FormatIdentifier VideoFrameSource::format_id() const
{
    std::string dec = decoder_name_;
    const size_t slash = dec.rfind('/');
    if (slash != std::string::npos) dec = dec.substr(slash + 1);
    return FormatIdentifier{std::string(to_string(demuxer_->container())), to_string(demuxer_->codec()), dec};
}

std::pair<ColorCharacteristics, ColorRange> VideoFrameSource::color_characteristics() const
{
    // color_characteristics_from_format (color.rs:36-50): the stream's codes, unspecified fields by height
    return {ColorCharacteristics::from_codes(fmt_.cp, fmt_.mc, fmt_.tc).or_(color_characteristics_fallback(fmt_.height)),
            fmt_.full_range ? ColorRange::Full : ColorRange::Limited};
}

} // namespace tm_host
