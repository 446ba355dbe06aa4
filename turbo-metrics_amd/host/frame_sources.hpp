// frame_sources.hpp -- CPU frame sources for the CLI: counterparts of the reference's FrameSource implementations
// (turbo-metrics/src/input_image.rs:91-229 for still images; its video sources are NVDEC demux/decode, which has no
// place on this hardware, so decoded video enters as Y4M / raw planar YUV and is handed to the engine as it is: planar 4:2:0
// through tm_engine_set_frame_i420, which converts it exactly like the NV12 / P016 surface of cudarse-video/src/dec.rs:299-403
// that the same samples would be repacked into).
//
//   ImageFrameSource   PNG (8/16-bit RGB, also Adam7), PPM P6 (8/16-bit), PFM "PF" (f32 RGB): like the reference,
//                      only RGB sample layouts are accepted (img.rs:17-37 is todo!() for anything else)
//   YuvStreamSource    YUV4MPEG2 (C420* 8-bit, C420p10 / p12 / p16) and headerless planar I420 / I420p10 with the size given on
//                      the command line -> HwFrame::Planar420; 10-bit streams are PACKED three samples to a word while they are copied into
//                      the page-locked ring (HwFrame::Planar420P10, tm_p10_pack_rows; TM_PACK10=0 hands them over as 16-bit words)
#pragma once
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <istream>
#include <memory>
#include <string>
#include <vector>

#include "turbo_metrics.hpp"

namespace tm_host {

constexpr size_t PROBE_LEN = 64; // input_image.rs:19

// what the first bytes look like; Unknown -> not an image (the CLI then tries the video path)
enum class ImageFormat { Unknown, PNG, PPM, PFM, JPEG, Other };
// needs at least PROBE_LEN bytes like ImageProbe::probe_image (input_image.rs:49-53), else throws "unexpected end of file"
ImageFormat probe_image(const unsigned char *start, size_t len);
bool can_decode(ImageFormat f);
const char *to_string(ImageFormat f);

struct CpuImg { // img.rs:40-48
    enum Sample { U8, U16, F32 } sample_type = U8;
    uint32_t width = 0, height = 0;
    std::vector<unsigned char> data; // packed RGB
};

// decoders (throw std::runtime_error with the reason)
CpuImg decode_png(const unsigned char *data, size_t len);                      // the first frame
std::vector<CpuImg> decode_png_frames(const unsigned char *data, size_t len); // every frame (animated PNG: composed onto the canvas)
CpuImg decode_pnm(const unsigned char *data, size_t len);

class ImageFrameSource : public FrameSource {
public:
    ImageFrameSource(std::vector<unsigned char> file, ImageFormat f);
    FormatIdentifier format_id() const override;
    uint32_t width() const override { return width_; }
    uint32_t height() const override { return height_; }
    std::pair<ColorCharacteristics, ColorRange> color_characteristics() const override;
    size_t frame_count() const override { return frames_.size(); }
    void skip_frames(uint32_t n) override;
    bool next_frame(HwFrame &out) override;

private:
    std::deque<CpuImg> frames_;
    CpuImg current_;
    ImageFormat format_;
    uint32_t width_ = 0, height_ = 0;
};

// planar 4:2:0 stream -> page-locked planar pictures
// A few persistent workers that split one picture: bringing it from the page cache into a page-locked surface is a pure
// memory copy (3 MB at 1080p, 25 MB at 4K 10-bit) and one thread per stream was what bounded the CLI end to end.
class RowWorkers {
public:
    explicit RowWorkers(unsigned n);
    ~RowWorkers();
    // fn(first, last) over [0, total), split into size() + 1 contiguous pieces; the caller works on the first one
    void run(size_t total, const std::function<void(size_t, size_t)> &fn);
    unsigned size() const { return (unsigned)th_.size(); }

private:
    void loop(unsigned idx);
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t, size_t)> *fn_ = nullptr;
    size_t total_ = 0;
    unsigned long long gen_ = 0;
    unsigned pending_ = 0;
    bool stop_ = false;
    std::exception_ptr err_; // first exception thrown by a piece of the current run() (rethrown by run() after every piece is done)
};

class YuvStreamSource : public FrameSource {
public:
    // takes ownership of `in` (closed with fclose unless it is stdin).  bits: 8, 10, 12 or 16.  header_frames: true = Y4M
    // ("FRAME...\n" before every picture)
    YuvStreamSource(FILE *in, bool y4m, uint32_t w, uint32_t h, int bits, ColorCharacteristics cc, ColorRange cr, size_t frame_count,
                    std::string codec);
    ~YuvStreamSource() override;
    FormatIdentifier format_id() const override;
    uint32_t width() const override { return w_; }
    uint32_t height() const override { return h_; }
    std::pair<ColorCharacteristics, ColorRange> color_characteristics() const override { return {cc_, cr_}; }
    size_t frame_count() const override { return frame_count_; }
    void skip_frames(uint32_t n) override;
    bool next_frame(HwFrame &out) override;
    bool skip_one() override; // (with the read-ahead running the picture is read like any other and not handed out)
    bool shardable() const override { return fd_ >= 0; }
    void set_lookahead(size_t frames) override;
    void set_readahead(bool on) override { if (ring_.empty()) readahead_ = on; }
    void prepare() override; // the whole ring, page-locked, before the first frame is asked for
    // bytes that were read from the stream before this source took it over (the format probe of a pipe)
    void set_prefix(std::vector<unsigned char> bytes);

private:
    size_t read_bytes(unsigned char *dst, size_t n);
    std::vector<unsigned char> prefix_;
    size_t prefix_pos_ = 0;
    bool read_picture(unsigned char *surface);
    FILE *in_;
    bool y4m_;
    uint32_t w_, h_;
    int bits_;
    ColorCharacteristics cc_;
    ColorRange cr_;
    size_t frame_count_;
    std::string codec_;
    size_t planar_bytes_ = 0;                   // one picture in the STREAM
    bool pack10_ = false;                       // 10-bit pictures are packed three samples to a word on their way into the ring
    size_t slot_bytes_ = 0, row_y_ = 0, row_c_ = 0; // one picture in the RING (packed: rows of whole 512-byte blocks), its row pitches
    void pack_picture(const unsigned char *planar, unsigned char *surface, size_t first_row, size_t last_row) const; // rows of Y, Cb, Cr counted through
    std::vector<unsigned char> planar_;         // pipes: scratch for a picture that is consumed but not handed out
    int fd_ = -1;                               // regular files: positioned reads (pread), split over the workers
    size_t file_size_ = 0, file_pos_ = 0;
    // ring of page-locked surfaces (tm_host_alloc): a frame stays valid for `lookahead` further next_frame calls, which
    // lets the engine pull it by asynchronous DMA; plain memory when page-locking fails
    // the ring of page-locked surfaces: slot 0 is allocated by the first next_frame, the others by a helper thread while the first
    // pictures are being read and computed (page-locking runs at ~3 GB/s: 0.28 s for the two rings of a 4K 10-bit pair at batch 8)
    std::vector<unsigned char *> ring_;
    std::vector<char> ring_pinned_;  // per slot: page-locked (tm_host_alloc) or plain memory (when page-locking fails: ring_failed_)
    unsigned char *ring_block_ = nullptr; // the whole ring as ONE page-locked allocation, slot after slot (round 6): pictures that follow each other in
                                          // the stream lie back to back, and the engine sends two to four of them up as one DMA (tm_engine.hip, queue_copy)
    size_t ring_pos_ = 0, lookahead_ = 1;
    size_t ring_ready_ = 0;          // slots usable so far (guarded by ring_m_)
    bool ring_failed_ = false;       // a page-locked allocation failed: the remaining slots come from pageable memory
    std::mutex ring_m_;
    std::condition_variable ring_cv_;
    std::thread ring_alloc_;
    void ensure_ring();                   // ring (alloc_ring) + readers
    void alloc_ring();
    unsigned reader_threads_ = 1;
    size_t ahead_ = 0;
    bool readers_started_ = false;
    unsigned char *ring_slot(size_t i);
    std::unique_ptr<RowWorkers> workers_; // created with the ring (pictures of 256 rows and more; synchronous mode)
    bool parse_frame_header();            // regular files: the FRAME line at file_pos_ (Y4M); false at the end of the stream
    // ---- read-ahead (regular files): a dispatcher walks the stream picture by picture and hands each to a pool of readers in pieces
    // of ~2 MB; next_frame only waits for its picture.  Picture p goes into ring slot p % ring: with a ring of lookahead + 1 + ahead
    // slots the readers may be `ahead` pictures in front of the next_frame call in progress (the slot of picture p - ring is free once
    // call p - ahead has started).  Whole pictures in flight instead of one picture split over N threads with a fork and a join per
    // picture: the per-picture thread wake-ups were a fifth of the time per 1080p picture.
    bool readahead_ = true;
    struct ReadAhead;
    std::unique_ptr<ReadAhead> ra_;
    void start_readahead(unsigned threads, size_t ahead);
    void stop_readahead();
};

struct SourceHints { // what a headerless stream cannot say about itself (CLI flags)
    uint32_t width = 0, height = 0;
    int bits = 8;
    int cp = 2, mc = 2, tc = 2; // H.273 codes, 2 = unspecified -> fallback by height (color.rs:51-78)
    bool full_range = false;
    bool force_raw = false;
};

// == create_source (turbo-metrics-cli/src/main.rs:172-209): probe the first PROBE_LEN bytes, pick a decoder.
// path "-" reads stdin.
std::unique_ptr<FrameSource> create_source(const std::string &path, const SourceHints &hints);

} // namespace tm_host
