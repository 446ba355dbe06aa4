// video_input.hpp -- compressed video input for the CLI: the counterpart of the reference's container / bitstream layer
// (crates/turbo-metrics/src/input_video.rs:19-441, crates/codec-bitstream/src/{lib,ivf,h264,av1,h262}.rs).
//
// What the reference does there: probe IVF / Matroska, demux the video track packet by packet (H.264 length-prefixed NAL
// units become Annex B, one NAL unit per call; AV1 / MPEG-2 packets whole), feed NVDEC's parser, and take size, bit depth and
// colour description from the decoder's format callback (color.rs:36-78).
// What exists on this platform: everything up to the decoder.  There is no hardware video decoder behind HIP on MI355X and
// none in this tree, so decoding is DELEGATED to an external decoder process (TM_DECODER, default `ffmpeg`): this layer demuxes
// itself, parses the sequence headers itself (H.264 SPS + VUI, AV1 sequence header OBU, MPEG-2 sequence header + display
// extension -- what cuvid's format callback would report), streams the elementary stream into the decoder's stdin and reads
// YUV4MPEG2 pictures from its stdout, which then take the planar-4:2:0 path of every other YUV input (frame_sources.hpp).
// Without a decoder program the source fails with a message that says so; nothing else in the CLI depends on it.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "frame_sources.hpp"

namespace tm_host {

enum class Codec { AV1, H264, MPEG2 };       // codec-bitstream/src/lib.rs:8-13
enum class Container { Mkv, Ivf };           // input_video.rs:18-22
const char *to_string(Codec c);              // "AV1" / "H264" / "MPEG2" (Display, lib.rs:26-40)
const char *to_string(Container c);          // "Mkv" / "Ivf" ({:?}, input_video.rs:375)

// what NVDEC's format callback hands the reference (CUVIDEOFORMAT): display size, bit depth, chroma format, video signal description
struct StreamFormat {
    bool valid = false;
    uint32_t width = 0, height = 0; // display size (after cropping)
    int bit_depth = 8;
    int chroma_format = 1;          // 0 mono, 1 4:2:0, 2 4:2:2, 3 4:4:4
    int cp = 2, mc = 2, tc = 2;     // H.273 codes, 2 = unspecified
    bool full_range = false;
};

// ---- bitstream headers -------------------------------------------------------------------------------------------------
// `nal`: one H.264 NAL unit starting at its header byte (no start code, emulation prevention bytes still in place)
StreamFormat h264_parse_sps(const uint8_t *nal, size_t len);
// first sequence header OBU found in `data` (a run of OBUs with size fields: an MKV CodecPrivate past its 4 bytes, an IVF packet)
StreamFormat av1_parse_sequence_header(const uint8_t *data, size_t len);
// MPEG-2 elementary stream bytes: sequence_header (00 00 01 B3) and, if present, the sequence extension / display extension
StreamFormat mpeg2_parse_sequence(const uint8_t *data, size_t len);
// == h264::avcc_extradata_to_annexb (h264.rs:168-197): SPS and PPS of an avcC record as Annex B; returns the NAL length size (0 = malformed)
size_t avcc_extradata_to_annexb(const std::vector<uint8_t> &codec_private, std::vector<uint8_t> &annexb);
// == h264::avcc_into_annexb (h264.rs:232-251): ONE length-prefixed NAL unit -> 00 00 00 01 + payload; returns the bytes consumed, 0 if incomplete
size_t avcc_into_annexb(const uint8_t *buf, size_t len, size_t nal_length_size, std::vector<uint8_t> &nalu);

struct IvfHeader { // ivf.rs:6-14
    uint8_t fourcc[4] = {0, 0, 0, 0};
    uint16_t w = 0, h = 0;
    uint32_t timebase_den = 0, timebase_num = 0, frames = 0;
};
// == ivf::read_header (ivf.rs:22-58); false when the signature is not DKIF or the bytes run out.  header_len: bytes to skip
bool ivf_read_header(const uint8_t *p, size_t n, IvfHeader &h, size_t &header_len);
bool codec_from_fourcc(const uint8_t fourcc[4], Codec &c);     // lib.rs:16-22: AV01, AVC1
bool codec_from_mkv_id(const std::string &id, Codec &c);       // input_video.rs:337-345

// ---- demuxers (trait Demuxer, input_video.rs:128-141) -------------------------------------------------------------------
class Demuxer {
public:
    virtual ~Demuxer() = default;
    virtual Container container() const = 0;
    virtual Codec codec() const = 0;
    virtual size_t frame_count() const = 0; // IVF: the header's count; MKV: 0 (input_video.rs:266-268)
    // out-of-band data that starts the elementary stream (avcC parameter sets as Annex B, AV1 sequence header, MPEG-2 CodecPrivate)
    virtual void init(std::vector<uint8_t> &out) = 0;
    // next piece of the video track as the reference feeds it to the parser: H.264 = ONE Annex B NAL unit, AV1 / MPEG-2 = one packet;
    // false at the end of the stream
    virtual bool demux(std::vector<uint8_t> &out) = 0;
};

// probe like VideoProbe::probe_file (input_video.rs:83-110): IVF by its header, else Matroska; nullptr + `why` when neither
// (why = the reference's ProbeError: "UnknownContainer", "IvfUnknownCodec(..)", "MKVNoVideo", "MkvUnknownCodec(..)").
// Takes ownership of `f` on success.
std::unique_ptr<Demuxer> probe_video(FILE *f, std::string &why);

// == VideoFrameSource (input_video.rs:347-441) with the decoder behind a pipe
// the decoder's arguments for "one picture out per picture decoded": `-fps_mode passthrough` (ffmpeg >= 5.1 or unknown), `-vsync passthrough` (older),
// or the tokens of TM_DECODER_ARGS
std::vector<std::string> decoder_sync_args(const std::string &prog);

class VideoFrameSource : public FrameSource {
public:
    VideoFrameSource(std::unique_ptr<Demuxer> demuxer, const SourceHints &hints);
    ~VideoFrameSource() override;
    FormatIdentifier format_id() const override;
    uint32_t width() const override { return inner_->width(); }
    uint32_t height() const override { return inner_->height(); }
    std::pair<ColorCharacteristics, ColorRange> color_characteristics() const override;
    size_t frame_count() const override { return demuxer_->frame_count(); }
    void skip_frames(uint32_t n) override { inner_->skip_frames(n); }
    bool next_frame(HwFrame &out) override { return inner_->next_frame(out); }
    bool skip_one() override { return inner_->skip_one(); }
    void set_lookahead(size_t frames) override { inner_->set_lookahead(frames); }
    void set_readahead(bool on) override { inner_->set_readahead(on); }
    void prepare() override { inner_->prepare(); }
    const StreamFormat &stream_format() const { return fmt_; }

private:
    void shutdown();
    std::unique_ptr<Demuxer> demuxer_;
    StreamFormat fmt_;
    std::string decoder_name_;
    int child_ = -1;
    std::thread feeder_;
    std::unique_ptr<FrameSource> inner_; // the decoder's YUV4MPEG2 output
};

// the YUV4MPEG2 stream that starts at the current position of `in` (shared with create_source): `head` = bytes already read
// from it; `seekable_file`: `in` is a regular file positioned right after `head`
std::unique_ptr<FrameSource> open_y4m_stream(FILE *in, std::string head, bool seekable_file, const SourceHints &hints, const std::string &what);

} // namespace tm_host
