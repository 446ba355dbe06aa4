// frame_sources.cpp -- see frame_sources.hpp
#include "frame_sources.hpp"
#include "video_input.hpp"
#include <cerrno>
#include <fcntl.h>
#include <unistd.h>

#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <sstream>

#include <sched.h>

namespace tm_host {

namespace {

[[noreturn]] void fail(const std::string &m) { throw std::runtime_error(m); }
// Dimension sanity shared by every reader: a corrupt header must end in an error, not in a 50 GB allocation (found by
// tools/fuzz_sources.py); 65 536 is far above anything the engine can hold and keeps w * h * 6 inside size_t arithmetic.
constexpr uint32_t MAX_DIM = 65536;
void check_dims(const char *what, uint64_t w, uint64_t h)
{
    if (w > MAX_DIM || h > MAX_DIM) fail(std::string(what) + ": implausible size " + std::to_string(w) + "x" + std::to_string(h));
}

uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

} // namespace

// ---- probing -------------------------------------------------------------------------------------------------------
ImageFormat probe_image(const unsigned char *s, size_t len)
{
    if (len < PROBE_LEN) fail("unexpected end of file"); // io::ErrorKind::UnexpectedEof, input_image.rs:51-53
    static const unsigned char png[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (!memcmp(s, png, 8)) return ImageFormat::PNG;
    if (s[0] == 'P' && s[1] == '6' && (s[2] == '\n' || s[2] == ' ' || s[2] == '\r' || s[2] == '\t')) return ImageFormat::PPM;
    if (s[0] == 'P' && s[1] == 'F' && (s[2] == '\n' || s[2] == ' ' || s[2] == '\r')) return ImageFormat::PFM;
    if (s[0] == 0xFF && s[1] == 0xD8 && s[2] == 0xFF) return ImageFormat::JPEG;
    if (!memcmp(s, "GIF8", 4) || !memcmp(s, "BM", 2) || (!memcmp(s, "RIFF", 4) && !memcmp(s + 8, "WEBP", 4)) ||
        !memcmp(s, "II*\0", 4) || !memcmp(s, "MM\0*", 4) || !memcmp(s, "qoif", 4))
        return ImageFormat::Other;
    return ImageFormat::Unknown;
}

bool can_decode(ImageFormat f) { return f == ImageFormat::PNG || f == ImageFormat::PPM || f == ImageFormat::PFM; }

const char *to_string(ImageFormat f)
{
    switch (f) {
    case ImageFormat::PNG: return "PNG";
    case ImageFormat::PPM: return "PPM";
    case ImageFormat::PFM: return "PFM";
    case ImageFormat::JPEG: return "JPEG";
    case ImageFormat::Other: return "Other";
    default: return "Unknown";
    }
}

// ---- PNG -----------------------------------------------------------------------------------------------------------
namespace {

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the per-scanline filters of one (sub)image in place; `raw` holds rows of 1 + stride bytes; returns packed rows
void unfilter(const unsigned char *raw, size_t rows, size_t stride, size_t bpp, std::vector<unsigned char> &out)
{
    out.assign(rows * stride, 0);
    std::vector<unsigned char> zero(stride, 0);
    for (size_t y = 0; y < rows; ++y) {
        const unsigned char ft = raw[y * (stride + 1)];
        const unsigned char *in = raw + y * (stride + 1) + 1;
        unsigned char *cur = out.data() + y * stride;
        const unsigned char *up = y ? cur - stride : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int v;
            switch (ft) {
            case 0: v = in[i]; break;
            case 1: v = in[i] + a; break;
            case 2: v = in[i] + b; break;
            case 3: v = in[i] + ((a + b) >> 1); break;
            case 4: v = in[i] + paeth(a, b, c); break;
            default: fail("PNG: invalid filter type");
            }
            cur[i] = (unsigned char)v;
        }
    }
}

} // namespace

namespace {

// one PNG (sub)image: zlib stream -> unfiltered, de-interlaced packed RGB rows of fw x fh pixels (samples in network order)
std::vector<unsigned char> png_inflate_image(const std::vector<unsigned char> &zdata, uint32_t fw, uint32_t fh, size_t bpp, int interlace)
{
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole[1] = {{0, 0, 1, 1}};
    const Pass *passes = interlace ? adam7 : whole;
    const int npass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int p = 0; p < npass; ++p) {
        const size_t pw = (fw > passes[p].x0) ? (fw - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const size_t ph = (fh > passes[p].y0) ? (fh - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (pw && ph) raw_size += ph * (1 + pw * bpp);
    }
    // deflate expands at most ~1032:1: a header that promises more than the compressed bytes can hold is corrupt (checked BEFORE
    // the allocation)
    if (raw_size / 1032 > zdata.size() + 1) fail("PNG: corrupt image data");
    std::vector<unsigned char> raw(raw_size);
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) fail("PNG: inflateInit failed");
        zs.next_in = const_cast<unsigned char *>(zdata.data()); zs.avail_in = (uInt)zdata.size();
        zs.next_out = raw.data(); zs.avail_out = (uInt)raw.size();
        if (zdata.size() > 0xFFFFFFFFu || raw.size() > 0xFFFFFFFFu) { inflateEnd(&zs); fail("PNG: image too large"); }
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = raw.size() - zs.avail_out;
        inflateEnd(&zs);
        if ((rc != Z_STREAM_END && rc != Z_BUF_ERROR && rc != Z_OK) || got != raw.size()) fail("PNG: corrupt image data");
    }
    std::vector<unsigned char> out((size_t)fw * fh * bpp, 0);
    size_t off = 0;
    std::vector<unsigned char> rows;
    for (int p = 0; p < npass; ++p) {
        const size_t pw = (fw > passes[p].x0) ? (fw - passes[p].x0 + passes[p].dx - 1) / passes[p].dx : 0;
        const size_t ph = (fh > passes[p].y0) ? (fh - passes[p].y0 + passes[p].dy - 1) / passes[p].dy : 0;
        if (!pw || !ph) continue;
        unfilter(raw.data() + off, ph, pw * bpp, bpp, rows);
        off += ph * (1 + pw * bpp);
        for (size_t y = 0; y < ph; ++y)
            for (size_t x = 0; x < pw; ++x)
                memcpy(out.data() + (((size_t)passes[p].y0 + y * passes[p].dy) * fw + passes[p].x0 + x * passes[p].dx) * bpp,
                       rows.data() + (y * pw + x) * bpp, bpp);
    }
    return out;
}

} // namespace

// Every frame of a PNG.  A plain PNG has one; an animated PNG (acTL / fcTL / fdAT, APNG 1.0) has the frames of its animation -- the
// reference hands every frame of a decoded image to the metric as a frame of the stream (input_image.rs:115-128: img.frames_ref()) --,
// each composed onto the canvas as the specification says: the frame's region is drawn at its offset (RGB has no alpha: blend_op OVER is
// SOURCE), the canvas after drawing is the frame handed out, and dispose_op tells what the region holds for the NEXT frame (NONE: this
// frame, BACKGROUND: zeros, PREVIOUS: what it held before; PREVIOUS on the first frame counts as BACKGROUND).  A default image that is not
// part of the animation (IDAT before the first fcTL) is skipped, as viewers of animated PNGs do.
std::vector<CpuImg> decode_png_frames(const unsigned char *d, size_t len)
{
    if (len < 8 + 25) fail("PNG: truncated");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    struct Frame { uint32_t w, h, x, y; int dispose, blend; bool is_default; std::vector<unsigned char> z; };
    std::vector<Frame> frames;
    std::vector<unsigned char> idat;
    bool animated = false, seen_idat = false, fctl_before_idat = false, end = false;
    uint32_t declared_frames = 0;
    Frame *cur = nullptr; // the frame that fdAT chunks (or IDAT, for a first frame that is the default image) add to
    while (!end && pos + 12 <= len) {
        const uint32_t clen = be32(d + pos);
        const unsigned char *type = d + pos + 4, *body = d + pos + 8;
        if (pos + 12 + (size_t)clen > len) fail("PNG: truncated chunk");
        if (!memcmp(type, "IHDR", 4)) {
            if (clen != 13) fail("PNG: bad IHDR");
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) fail("PNG: unknown compression/filter method");
            check_dims("PNG", w, h);
        } else if (!memcmp(type, "acTL", 4) && !seen_idat) {
            if (clen != 8) fail("PNG: bad acTL");
            declared_frames = be32(body);
            if (declared_frames == 0 || declared_frames > (1u << 20)) fail("PNG: implausible number of animation frames");
            animated = true;
        } else if (!memcmp(type, "fcTL", 4) && animated) {
            if (clen != 26) fail("PNG: bad fcTL");
            if (ctype < 0) fail("PNG: fcTL before IHDR");
            Frame f{be32(body + 4), be32(body + 8), be32(body + 12), be32(body + 16), body[24], body[25], !seen_idat, {}};
            if (f.w == 0 || f.h == 0 || (uint64_t)f.x + f.w > w || (uint64_t)f.y + f.h > h) fail("PNG: animation frame outside the canvas");
            if (f.dispose > 2 || f.blend > 1) fail("PNG: unknown dispose / blend operation");
            if (!seen_idat) { if (fctl_before_idat) fail("PNG: two fcTL chunks before IDAT"); fctl_before_idat = true; if (f.w != w || f.h != h || f.x || f.y) fail("PNG: the first frame must cover the canvas"); }
            if (frames.size() >= declared_frames) fail("PNG: more animation frames than acTL declares");
            frames.push_back(std::move(f));
            cur = &frames.back();
        } else if (!memcmp(type, "IDAT", 4)) {
            seen_idat = true;
            idat.insert(idat.end(), body, body + clen);
        } else if (!memcmp(type, "fdAT", 4) && animated) {
            if (clen < 4 || !cur || cur->is_default) fail("PNG: fdAT without its fcTL");
            cur->z.insert(cur->z.end(), body + 4, body + clen);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)clen;
    }
    if (ctype < 0 || w == 0 || h == 0) fail("PNG: no IHDR");
    // the reference accepts RGB sample layouts only (turbo-metrics/src/img.rs:17-37: anything else is todo!())
    if (ctype != 2) fail("not implemented: PNG colour type " + std::to_string(ctype) + " (only RGB is supported, as in the reference)");
    if (depth != 8 && depth != 16) fail("PNG: RGB must be 8 or 16 bits per sample");
    if (interlace > 1) fail("PNG: unknown interlace method");
    const size_t bpp = 3 * (size_t)depth / 8;
    if (!animated || frames.empty()) { // a plain PNG (or an acTL without any frame: the default image)
        frames.clear();
        frames.push_back(Frame{w, h, 0, 0, 0, 0, true, {}});
    }
    if ((uint64_t)w * h * bpp * frames.size() > ((uint64_t)1 << 34)) fail("PNG: animation too large");
    std::vector<CpuImg> out;
    std::vector<unsigned char> canvas((size_t)w * h * bpp, 0), before;
    for (size_t k = 0; k < frames.size(); ++k) {
        const Frame &f = frames[k];
        const std::vector<unsigned char> px = png_inflate_image(f.is_default ? idat : f.z, f.w, f.h, bpp, interlace);
        const int dispose = (k == 0 && f.dispose == 2) ? 1 : f.dispose;
        if (dispose == 2) before = canvas;
        for (uint32_t y = 0; y < f.h; ++y)
            memcpy(canvas.data() + (((size_t)f.y + y) * w + f.x) * bpp, px.data() + (size_t)y * f.w * bpp, (size_t)f.w * bpp);
        CpuImg img;
        img.width = w; img.height = h;
        img.sample_type = depth == 8 ? CpuImg::U8 : CpuImg::U16;
        img.data = canvas;
        if (depth == 16) // network order -> host order
            for (size_t i = 0; i + 1 < img.data.size(); i += 2) std::swap(img.data[i], img.data[i + 1]);
        out.push_back(std::move(img));
        if (dispose == 1)
            for (uint32_t y = 0; y < f.h; ++y) memset(canvas.data() + (((size_t)f.y + y) * w + f.x) * bpp, 0, (size_t)f.w * bpp);
        else if (dispose == 2) canvas = before;
    }
    return out;
}

CpuImg decode_png(const unsigned char *d, size_t len) { return std::move(decode_png_frames(d, len).front()); }

// ---- PPM (P6) / PFM (PF) --------------------------------------------------------------------------------------------
CpuImg decode_pnm(const unsigned char *d, size_t len)
{
    size_t pos = 2;
    auto token = [&]() {
        for (;;) {
            while (pos < len && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) ++pos;
            if (pos < len && d[pos] == '#') { while (pos < len && d[pos] != '\n') ++pos; continue; }
            break;
        }
        std::string t;
        while (pos < len && !(d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) t.push_back((char)d[pos++]);
        if (t.empty()) fail("PNM: truncated header");
        return t;
    };
    const bool pfm = d[1] == 'F';
    CpuImg img;
    const unsigned long pw = std::stoul(token()), ph = std::stoul(token());
    check_dims("PNM", pw, ph);
    img.width = (uint32_t)pw;
    img.height = (uint32_t)ph;
    const std::string third = token();
    ++pos; // the single whitespace byte after the header
    if (img.width == 0 || img.height == 0) fail("PNM: empty image");
    const size_t n = (size_t)img.width * img.height * 3;
    if (pfm) {
        const double scale = std::stod(third);
        if (pos + n * 4 > len) fail("PFM: truncated");
        img.sample_type = CpuImg::F32;
        img.data.resize(n * 4);
        const size_t row = (size_t)img.width * 12;
        for (uint32_t y = 0; y < img.height; ++y) { // PFM rows run bottom to top
            const unsigned char *src = d + pos + (size_t)(img.height - 1 - y) * row;
            unsigned char *dst = img.data.data() + (size_t)y * row;
            if (scale < 0) memcpy(dst, src, row);
            else for (size_t i = 0; i < row; i += 4) { dst[i] = src[i + 3]; dst[i + 1] = src[i + 2]; dst[i + 2] = src[i + 1]; dst[i + 3] = src[i]; }
        }
    } else {
        const unsigned long maxval = std::stoul(third);
        if (maxval != 255 && maxval != 65535) fail("PPM: maxval must be 255 or 65535");
        const size_t bps = maxval == 255 ? 1 : 2;
        if (pos + n * bps > len) fail("PPM: truncated");
        img.sample_type = bps == 1 ? CpuImg::U8 : CpuImg::U16;
        img.data.assign(d + pos, d + pos + n * bps);
        if (bps == 2)
            for (size_t i = 0; i + 1 < img.data.size(); i += 2) std::swap(img.data[i], img.data[i + 1]);
    }
    return img;
}

// ---- ImageFrameSource ------------------------------------------------------------------------------------------------
ImageFrameSource::ImageFrameSource(std::vector<unsigned char> file, ImageFormat f) : format_(f)
{
    // every frame of the decoded image becomes a frame of the stream (input_image.rs:115-128); PPM / PFM hold one
    if (f == ImageFormat::PNG) { for (CpuImg &img : decode_png_frames(file.data(), file.size())) frames_.push_back(std::move(img)); }
    else frames_.push_back(decode_pnm(file.data(), file.size()));
    width_ = frames_.front().width; height_ = frames_.front().height;
}

FormatIdentifier ImageFrameSource::format_id() const { return FormatIdentifier{std::nullopt, to_string(format_), "turbo-metrics-hip"}; }

std::pair<ColorCharacteristics, ColorRange> ImageFrameSource::color_characteristics() const
{
    // input_image.rs:183-194: image colour metadata is not read; unused on the image path
    ColorCharacteristics c;
    c.cp = ColourPrimaries::BT709; c.mc = MatrixCoefficients::BT709; c.tc = TransferCharacteristic::BT709;
    return {c, ColorRange::Full};
}

void ImageFrameSource::skip_frames(uint32_t n)
{
    for (uint32_t i = 0; i < n && !frames_.empty(); ++i) frames_.pop_front();
}

bool ImageFrameSource::next_frame(HwFrame &out)
{
    if (frames_.empty()) return false;
    current_ = std::move(frames_.front());
    frames_.pop_front();
    out = HwFrame{};
    const size_t bps = current_.sample_type == CpuImg::U8 ? 1 : (current_.sample_type == CpuImg::U16 ? 2 : 4);
    out.kind = current_.sample_type == CpuImg::U8 ? HwFrame::Npp8 : (current_.sample_type == CpuImg::U16 ? HwFrame::Npp16 : HwFrame::Npp32);
    out.data = current_.data.data();
    out.pitch = (size_t)current_.width * 3 * bps;
    return true;
}

// pread the whole range or fail
static void pread_all(int fd, unsigned char *dst, size_t n, size_t off)
{
    while (n > 0) {
        const ssize_t got = pread(fd, dst, n, (off_t)off);
        if (got < 0) { if (errno == EINTR) continue; fail(std::string("read error: ") + strerror(errno)); }
        if (got == 0) fail("truncated picture in the YUV stream");
        dst += got; off += (size_t)got; n -= (size_t)got;
    }
}

static std::atomic<unsigned> g_concurrent_streams{2};
void set_concurrent_streams(unsigned n) { g_concurrent_streams.store(n < 1 ? 1 : n); }

unsigned effective_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    // cgroup v2: "max 100000" or "<quota> <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = no limit)
    long long quota = -1, period = 0;
    {
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string q;
        if (f >> q >> period && q != "max") quota = atoll(q.c_str());
    }
    if (quota <= 0) {
        std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        if (!(fq >> quota) || !(fp >> period)) quota = -1;
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

// ---- planar 4:2:0 streams ---------------------------------------------------------------------------------------------
struct YuvStreamSource::ReadAhead {
    struct Piece { size_t pic, off, len; unsigned char *dst; uint32_t rows, width; size_t dst_pitch; }; // rows > 0: `rows` rows of `width` 10-bit samples, packed into dst
    std::vector<std::thread> pool;
    std::thread dispatcher;
    std::mutex m;
    std::condition_variable cv_job, cv_done, cv_room;
    std::deque<Piece> q;
    std::vector<unsigned> left;   // [slot]: pieces of the picture in it that are still being read
    std::vector<size_t> pic_of;   // [slot]: the picture the slot holds or is being filled with (SIZE_MAX: none yet)
    size_t ahead = 1;
    size_t entered = 0;           // next_frame calls started so far
    size_t eof_pic = SIZE_MAX;    // index of the first picture that does not exist
    size_t err_pic = SIZE_MAX;    // first picture that could not be read (err: why); pictures before it are still delivered
    std::exception_ptr err;
    std::atomic<bool> quit{false};
};

YuvStreamSource::YuvStreamSource(FILE *in, bool y4m, uint32_t w, uint32_t h, int bits, ColorCharacteristics cc, ColorRange cr,
                                 size_t frame_count, std::string codec)
    : in_(in), y4m_(y4m), w_(w), h_(h), bits_(bits), cc_(cc), cr_(cr), frame_count_(frame_count), codec_(std::move(codec))
{
    const size_t bps = bits_ > 8 ? 2 : 1, cw = (w_ + 1) / 2, ch = (h_ + 1) / 2;
    planar_bytes_ = ((size_t)w_ * h_ + 2 * cw * ch) * bps;
    // 10-bit pictures cross PCIe packed: 10.7 instead of 16 bits per sample (37.5 % of a yuv420p10 upload is zeros, and 4K end to end is
    // bound by that link).  The readers pack while they copy a picture into the page-locked ring.
    const char *pk = getenv("TM_PACK10");
    pack10_ = bits_ == 10 && !(pk && atoi(pk) == 0);
    row_y_ = pack10_ ? tm_p10_row_bytes(w_) : (size_t)w_ * bps;
    row_c_ = pack10_ ? tm_p10_row_bytes((uint32_t)cw) : cw * bps;
    slot_bytes_ = row_y_ * h_ + 2 * row_c_ * ch;
    // a regular file is read with pread(), every worker its own byte range, straight from the page cache into the page-locked
    // ring (a mapping of the file costs a minor page fault per 4 KB on first touch: 1.5 M faults for a 6-GB clip, which was what
    // bounded 4K streams)
    struct stat st;
    const int fd = fileno(in_);
    const long at = ftell(in_);
    if (fd >= 0 && at >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        fd_ = fd; file_size_ = (size_t)st.st_size; file_pos_ = (size_t)at;
        posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    }
}

RowWorkers::RowWorkers(unsigned n)
{
    for (unsigned i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
}

RowWorkers::~RowWorkers()
{
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
}

void RowWorkers::loop(unsigned idx)
{
    unsigned long long seen = 0;
    for (;;) {
        const std::function<void(size_t, size_t)> *fn;
        size_t total;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            fn = fn_; total = total_;
        }
        const size_t parts = th_.size() + 1, lo = total * (idx + 1) / parts, hi = total * (idx + 2) / parts;
        std::exception_ptr ex;
        try {
            if (hi > lo) (*fn)(lo, hi);
        } catch (...) { ex = std::current_exception(); } // a short read / EIO in a piece: reported by run(), never std::terminate
        {
            std::lock_guard<std::mutex> lk(m_);
            if (ex && !err_) err_ = ex;
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
}

void RowWorkers::run(size_t total, const std::function<void(size_t, size_t)> &fn)
{
    {
        std::lock_guard<std::mutex> lk(m_);
        fn_ = &fn; total_ = total; pending_ = (unsigned)th_.size(); ++gen_; err_ = nullptr;
    }
    cv_.notify_all();
    const size_t hi = total / (th_.size() + 1);
    std::exception_ptr mine;
    try {
        if (hi > 0) fn(0, hi);
    } catch (...) { mine = std::current_exception(); }
    // always wait: the workers hold a pointer to `fn` and write into the caller's buffer until their pieces are done
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    std::exception_ptr first = mine ? mine : err_;
    err_ = nullptr;
    lk.unlock();
    if (first) std::rethrow_exception(first);
}

YuvStreamSource::~YuvStreamSource()
{
    stop_readahead();
    if (ring_alloc_.joinable()) ring_alloc_.join();
    workers_.reset();
    if (in_ && in_ != stdin) fclose(in_);
    if (ring_block_) tm_host_free(ring_block_);
    else
        for (size_t i = 0; i < ring_.size(); ++i) {
            if (!ring_[i]) continue;
            if (ring_pinned_[i]) tm_host_free(ring_[i]);
            else free(ring_[i]);
        }
}

void YuvStreamSource::stop_readahead()
{
    if (!ra_) return;
    {
        std::lock_guard<std::mutex> g(ra_->m);
        ra_->quit = true;
    }
    ra_->cv_job.notify_all(); ra_->cv_room.notify_all(); ra_->cv_done.notify_all();
    { std::lock_guard<std::mutex> g(ring_m_); } // (the dispatcher may be waiting for a ring slot: it looks at quit under this lock)
    ring_cv_.notify_all();
    if (ra_->dispatcher.joinable()) ra_->dispatcher.join();
    for (auto &t : ra_->pool) t.join();
    ra_.reset();
}

void YuvStreamSource::set_prefix(std::vector<unsigned char> bytes)
{
    prefix_ = std::move(bytes);
    prefix_pos_ = 0;
    if (!prefix_.empty()) fd_ = -1; // a stream whose first bytes were consumed by the probe is read sequentially
}

size_t YuvStreamSource::read_bytes(unsigned char *dst, size_t n)
{
    size_t done = 0;
    if (prefix_pos_ < prefix_.size()) {
        done = std::min(n, prefix_.size() - prefix_pos_);
        memcpy(dst, prefix_.data() + prefix_pos_, done);
        prefix_pos_ += done;
    }
    if (done < n) done += fread(dst + done, 1, n - done, in_);
    return done;
}

void YuvStreamSource::set_lookahead(size_t frames)
{
    if (ring_.empty()) lookahead_ = frames ? frames : 1;
}

void YuvStreamSource::prepare()
{
    alloc_ring();
    if (ring_alloc_.joinable()) ring_alloc_.join(); // every slot page-locked (or known not to be) before the caller's clock starts
}

void YuvStreamSource::ensure_ring()
{
    alloc_ring();
    if (readers_started_) return;
    readers_started_ = true;
    // the readers start with the first frame that is asked for: skip_frames (sequential, before) has moved the file position by then
    if (ahead_ > 0) start_readahead(reader_threads_, ahead_);
    else if (h_ >= 256 && reader_threads_ > 1 && fd_ >= 0) workers_ = std::make_unique<RowWorkers>(std::min(reader_threads_, 32u) - 1);
}

void YuvStreamSource::alloc_ring()
{
    if (!ring_.empty()) return;
    // reader threads per stream (the reference and the distorted stream are read at the same time): what the process may really use
    // (effective_cpus: hardware threads, affinity mask, cgroup quota) less four for the main thread, helpers and the HIP runtime, a third
    // of the rest (two streams, and head room under the quota); one thread per 384 KB of picture at most (3 MB at 1080p, 25 MB at 4K 10-bit), at most 16.  TM_READER_THREADS overrides; 1 = serial.
    // Measured on a host of 256 hardware threads (rounds 2-3): 1080p 4.97 k pairs/s with 8 threads per stream, 4.45 k with 16; 4K
    // 10-bit 640 / 822 / 536 pairs/s with 8 / 16 / 32 -- that host's container is limited to 16 CPUs of time, which is where the
    // optimum and its instability from box to box came from.
    const char *env = getenv("TM_READER_THREADS");
    const unsigned cpus = effective_cpus();
    const unsigned by_size = (unsigned)std::min<size_t>(16, std::max<size_t>(2, planar_bytes_ / (384u << 10)));
    // (round 4, reader alone on a box of this pool -- 16 CPUs of quota -- two streams at once, file in tmpfs -> page-locked ring: 1 / 2 / 4 / 7 /
    // 12 / 16 threads per stream = 9 / 12 / 17 / 20 / 11 / 18 GB/s per stream at 1080p, 6 / 8 / 17 / 15 / 12 / 16 at 4K: a thread copies
    // 6-9 GB/s, the pair of streams tops out at 35-40 GB/s, and more busy threads than the quota get the group throttled:
    // profiles/r04h_read_probe.log) -> a third of the CPUs left after main thread, dispatchers, ring helper and the HIP runtime
    // (streams: how many sources this process reads at the same time -- two, or two per device with `--devices N`)
    const unsigned streams = g_concurrent_streams.load();
    // (round 6, 10-bit streams packed on the way in: a reader now also packs what it reads, and writes a third less -- on the same 16-CPU box
    // 4 / 6 / 8 threads per stream = 1.13-1.38 / 1.41-1.51 / 1.41-1.51 k pairs/s of 4K, from 6 on the loop waits for the uploads instead of the
    // sources: profiles/r06h_pack_piece_ab.log -> half of the CPUs left after the helpers, per stream)
    const unsigned want = env ? (unsigned)std::max(1, atoi(env)) : std::min(by_size, std::max(1u, (cpus > 4 ? cpus - 4 : 1) / (pack10_ ? streams : streams + 1)));
    const bool ra = readahead_ && fd_ >= 0 && h_ >= 64;
    // read-ahead: enough pictures in flight to keep `want` readers busy with pieces of ~2 MB, at most 256 MB of them
    const size_t pieces = std::max<size_t>(1, (planar_bytes_ + ((size_t)2 << 20) - 1) / ((size_t)2 << 20));
    const size_t ahead = ra ? std::max<size_t>(2, std::min<size_t>((2 * want + pieces - 1) / pieces, std::max<size_t>(2, ((size_t)256 << 20) / planar_bytes_))) : 0;
    const size_t n = lookahead_ + 1 + ahead;
    ring_.assign(n, nullptr);
    ring_pinned_.assign(n, 0);
    // the whole ring in one page-locked piece when the pictures are small (1080p: 25 MB, ~10 ms of page-locking; larger rings -- 4K -- are built
    // slot by slot on a helper thread while the first pictures are read: their pictures are a DMA each anyway)
    if (n * slot_bytes_ <= ((size_t)64 << 20) && (ring_block_ = (unsigned char *)tm_host_alloc(n * slot_bytes_)) != nullptr) {
        for (size_t i = 0; i < n; ++i) { ring_[i] = ring_block_ + i * slot_bytes_; ring_pinned_[i] = 1; }
        ring_ready_ = n;
        reader_threads_ = want;
        ahead_ = ahead;
        return;
    }
    // slot 0 here ...
    ring_[0] = (unsigned char *)tm_host_alloc(slot_bytes_);
    ring_pinned_[0] = ring_[0] != nullptr;
    if (!ring_pinned_[0]) { // no page-locked memory at all: the whole ring is plain memory (the engine then copies synchronously)
        for (size_t i = 0; i < n; ++i) {
            ring_[i] = (unsigned char *)calloc(1, slot_bytes_);
            if (!ring_[i]) fail("out of memory for the frame ring");
        }
        ring_ready_ = n;
    } else {
        ring_ready_ = 1;
        // ... the others on a helper thread, in the order the readers will want them.  When page-locking fails part of the way
        // (the page-lock limit: 4K 10-bit rings at a large --batch), the REST of the ring is plain memory -- a frame says per slot
        // whether it is page-locked (HwFrame::pinned), and a pageable frame is copied synchronously by the engine
        ring_alloc_ = std::thread([this, n] {
            bool pinned = true;
            for (size_t i = 1; i < n; ++i) {
                unsigned char *p = pinned ? (unsigned char *)tm_host_alloc(slot_bytes_) : nullptr;
                if (!p) { pinned = false; p = (unsigned char *)calloc(1, slot_bytes_); }
                std::lock_guard<std::mutex> g(ring_m_);
                if (!p) { ring_failed_ = true; ring_cv_.notify_all(); return; } // out of plain memory too
                ring_[i] = p;
                ring_pinned_[i] = pinned;
                ring_ready_ = i + 1;
                ring_cv_.notify_all();
            }
        });
    }
    reader_threads_ = want;
    ahead_ = ahead;
}

unsigned char *YuvStreamSource::ring_slot(size_t i)
{
    std::unique_lock<std::mutex> lk(ring_m_);
    ring_cv_.wait(lk, [&] { return ring_ready_ > i || ring_failed_ || (ra_ && ra_->quit); });
    if (ring_ready_ <= i) fail("out of memory for the frame ring");
    return ring_[i];
}

// the FRAME line of a Y4M picture at file_pos_ (regular files); false at the end of the stream; throws on a malformed header
bool YuvStreamSource::parse_frame_header()
{
    if (file_pos_ >= file_size_) return false;
    if (y4m_) {
        unsigned char head[256];
        const size_t n = std::min<size_t>(file_size_ - file_pos_, sizeof head);
        pread_all(fd_, head, n, file_pos_);
        if (n < 6 || memcmp(head, "FRAME", 5)) fail("Y4M: expected a FRAME header");
        const void *nl = memchr(head, '\n', n);
        if (!nl) fail("Y4M: truncated FRAME header");
        file_pos_ += (size_t)((const unsigned char *)nl - head) + 1;
    }
    if (file_size_ - file_pos_ < planar_bytes_) {
        if (!y4m_) return false; // a trailing partial picture of a raw stream is ignored
        fail("truncated picture in the YUV stream");
    }
    return true;
}

void YuvStreamSource::start_readahead(unsigned threads, size_t ahead)
{
    ra_ = std::make_unique<ReadAhead>();
    ReadAhead &R = *ra_;
    R.ahead = ahead;
    R.left.assign(ring_.size(), 0);
    R.pic_of.assign(ring_.size(), SIZE_MAX);
    const size_t piece = (size_t)2 << 20;
    R.dispatcher = std::thread([this, &R, piece] {
        for (size_t p = 0;; ++p) {
            {
                std::unique_lock<std::mutex> lk(R.m);
                R.cv_room.wait(lk, [&] { return R.quit || p < R.entered + R.ahead; });
                if (R.quit) return;
            }
            const size_t slot = p % ring_.size();
            unsigned char *dst = nullptr;
            size_t at = 0;
            bool more = false;
            std::exception_ptr ex;
            try {
                dst = ring_slot(slot);
                more = parse_frame_header();
                at = file_pos_;
                if (more) file_pos_ += planar_bytes_;
            } catch (...) { ex = std::current_exception(); }
            std::lock_guard<std::mutex> g(R.m);
            if (R.quit) return;
            if (ex) { if (p < R.err_pic) { R.err_pic = p; R.err = ex; } R.cv_done.notify_all(); return; }
            if (!more) { R.eof_pic = p; R.cv_done.notify_all(); return; }
            R.pic_of[slot] = p;
            if (pack10_) { // whole rows of one plane per piece, ~1 MB of the stream each: the 16-bit rows pass through a reader's caches, not through memory
                const size_t cw = (w_ + 1) / 2, ch = (h_ + 1) / 2;
                struct Plane { size_t off, rows, width, pitch; unsigned char *dst; } planes[3] = {
                    {at, h_, w_, row_y_, dst}, {at + (size_t)w_ * h_ * 2, ch, cw, row_c_, dst + row_y_ * h_}, {at + ((size_t)w_ * h_ + cw * ch) * 2, ch, cw, row_c_, dst + row_y_ * h_ + row_c_ * ch}};
                unsigned n = 0;
                for (const Plane &pl : planes) {
                    // (pieces of 64 KB / 128 KB / 256 KB / 512 KB / 1 MB / 2 MB: 1.09 / 1.2 / 1.1-1.4 / 1.3 / 1.3-1.4 / 1.25-1.4 k pairs/s of 4K with four readers per
                    // stream -- small pieces pay for the queue, large ones leave the cache: 1 MB; TM_PACK_PIECE_KB overrides for a measurement)
                    static const size_t piece_kb = [] { const char *e = getenv("TM_PACK_PIECE_KB"); const int v = e ? atoi(e) : 0; return (size_t)(v >= 16 && v <= 8192 ? v : 1024); }();
                    const size_t per = std::max<size_t>(1, (piece_kb << 10) / (pl.width * 2));
                    for (size_t r = 0; r < pl.rows; r += per) {
                        const size_t k = std::min(per, pl.rows - r);
                        R.q.push_back(ReadAhead::Piece{p, pl.off + r * pl.width * 2, k * pl.width * 2, pl.dst + r * pl.pitch, (uint32_t)k, (uint32_t)pl.width, pl.pitch});
                        ++n;
                    }
                }
                R.left[slot] = n;
            } else {
                const size_t npieces = (planar_bytes_ + piece - 1) / piece;
                R.left[slot] = (unsigned)npieces;
                for (size_t i = 0; i < npieces; ++i) {
                    const size_t first = i * piece, len = std::min(piece, planar_bytes_ - first);
                    R.q.push_back(ReadAhead::Piece{p, at + first, len, dst + first, 0, 0, 0});
                }
            }
            R.cv_job.notify_all();
        }
    });
    for (unsigned t = 0; t < std::max(1u, threads); ++t)
        R.pool.emplace_back([this, &R] {
            for (;;) {
                ReadAhead::Piece pc;
                {
                    std::unique_lock<std::mutex> lk(R.m);
                    R.cv_job.wait(lk, [&] { return R.quit || !R.q.empty(); });
                    if (R.quit) return;
                    pc = R.q.front();
                    R.q.pop_front();
                }
                std::exception_ptr ex;
                try {
                    if (pc.rows) { // 10-bit rows: stream -> this thread's scratch -> packed into the ring
                        static thread_local std::vector<unsigned char> scratch;
                        if (scratch.size() < pc.len) scratch.resize(pc.len);
                        pread_all(fd_, scratch.data(), pc.len, pc.off);
                        tm_p10_pack_rows(scratch.data(), (size_t)pc.width * 2, pc.width, pc.rows, pc.dst, pc.dst_pitch);
                    } else pread_all(fd_, pc.dst, pc.len, pc.off);
                } catch (...) { ex = std::current_exception(); }
                std::lock_guard<std::mutex> g(R.m);
                if (ex && pc.pic < R.err_pic) { R.err_pic = pc.pic; R.err = ex; }
                if (--R.left[pc.pic % ring_.size()] == 0 || ex) R.cv_done.notify_all();
            }
        });
}

FormatIdentifier YuvStreamSource::format_id() const
{
    return FormatIdentifier{y4m_ ? std::optional<std::string>("Y4M") : std::nullopt, codec_, "turbo-metrics-hip"};
}

// next picture -> `surface` (planar_bytes_ bytes: Y, Cb, Cr planes back to back, as in the stream); nullptr: consume it only
bool YuvStreamSource::read_picture(unsigned char *surface)
{
    if (fd_ >= 0) { // regular file: positioned reads, split over the workers
        if (!parse_frame_header()) return false;
        const size_t at = file_pos_;
        file_pos_ += planar_bytes_;
        if (!surface) return true;
        unsigned char *into = surface;
        if (pack10_) { // the 16-bit picture goes through a scratch copy, then row by row into the packed surface
            if (planar_.size() != planar_bytes_) planar_.resize(planar_bytes_);
            into = planar_.data();
        }
        const std::function<void(size_t, size_t)> piece = [&](size_t first, size_t last) { pread_all(fd_, into + first, last - first, at + first); };
        if (workers_) workers_->run(planar_bytes_, piece);
        else piece(0, planar_bytes_);
        if (pack10_) {
            const size_t rows = (size_t)h_ + 2 * ((h_ + 1) / 2);
            const std::function<void(size_t, size_t)> pack = [&](size_t first, size_t last) { pack_picture(planar_.data(), surface, first, last); };
            if (workers_) workers_->run(rows, pack);
            else pack(0, rows);
        }
        return true;
    }
    // a pipe: sequential reads, straight into the surface (or into a scratch picture when it is only consumed)
    if (y4m_) {
        unsigned char tag[6];
        const size_t got = read_bytes(tag, 5);
        if (got == 0) return false;
        if (got != 5 || memcmp(tag, "FRAME", 5)) fail("Y4M: expected a FRAME header");
        unsigned char c;
        do {
            if (read_bytes(&c, 1) != 1) fail("Y4M: truncated FRAME header");
        } while (c != '\n');
    }
    unsigned char *into = surface;
    if (!surface || pack10_) {
        if (planar_.size() != planar_bytes_) planar_.resize(planar_bytes_);
        into = planar_.data();
    }
    const size_t got = read_bytes(into, planar_bytes_);
    if (got == 0 && !y4m_) return false;
    if (got != planar_bytes_) {
        if (!y4m_) return false; // a trailing partial picture of a raw stream is ignored
        fail("truncated picture in the YUV stream");
    }
    if (surface && pack10_) pack_picture(planar_.data(), surface, 0, (size_t)h_ + 2 * ((h_ + 1) / 2));
    return true;
}

// rows [first_row, last_row) of a 10-bit picture, counted through its Y, Cb and Cr planes: from the stream's 16-bit rows to packed rows
void YuvStreamSource::pack_picture(const unsigned char *planar, unsigned char *surface, size_t first_row, size_t last_row) const
{
    const size_t cw = (w_ + 1) / 2, ch = (h_ + 1) / 2;
    for (size_t r = first_row; r < last_row; ++r) {
        if (r < h_) tm_p10_pack_rows(planar + r * w_ * 2, (size_t)w_ * 2, w_, 1, surface + r * row_y_, row_y_);
        else {
            const size_t plane = r - h_ < ch ? 0 : 1, cr = r - h_ - plane * ch;
            tm_p10_pack_rows(planar + ((size_t)w_ * h_ + (plane * ch + cr) * cw) * 2, cw * 2, (uint32_t)cw, 1, surface + row_y_ * h_ + (plane * ch + cr) * row_c_, row_c_);
        }
    }
}

bool YuvStreamSource::skip_one()
{
    if (ra_) { HwFrame f; return next_frame(f); }
    return read_picture(nullptr);
}

void YuvStreamSource::skip_frames(uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
        if (!read_picture(nullptr)) break;
}

bool YuvStreamSource::next_frame(HwFrame &out)
{
    ensure_ring();
    unsigned char *surface = nullptr;
    size_t slot = ring_pos_;
    if (ra_) { // the picture is (being) read by the pool: wait for it
        ReadAhead &R = *ra_;
        std::unique_lock<std::mutex> lk(R.m);
        const size_t k = R.entered; // this call's picture
        slot = k % ring_.size();
        R.entered = k + 1;
        R.cv_room.notify_all();
        R.cv_done.wait(lk, [&] { return (R.pic_of[slot] == k && R.left[slot] == 0) || R.eof_pic <= k || R.err_pic <= k; });
        if (R.err_pic <= k) std::rethrow_exception(R.err);
        if (R.eof_pic <= k) return false;
        lk.unlock();
        surface = ring_slot(slot);
    } else {
        surface = ring_slot(slot);
        if (!read_picture(surface)) return false;
        ring_pos_ = (ring_pos_ + 1) % ring_.size();
    }
    const size_t ch = (h_ + 1) / 2;
    out = HwFrame{};
    out.kind = pack10_ ? HwFrame::Planar420P10 : HwFrame::Planar420;
    out.data = surface;
    out.u = surface + row_y_ * h_;
    out.v = surface + row_y_ * h_ + row_c_ * ch;
    out.pitch = row_y_;
    out.pitch_uv = row_c_;
    out.bits = bits_;
    out.pinned = ring_pinned_[slot] != 0;
    return true;
}

// ---- create_source ---------------------------------------------------------------------------------------------------
namespace {

long file_size_or_zero(FILE *f)
{
    const long at = ftell(f);
    if (at < 0 || fseek(f, 0, SEEK_END) != 0) return 0;
    const long end = ftell(f);
    fseek(f, at, SEEK_SET);
    return end;
}

} // namespace

std::unique_ptr<FrameSource> create_source(const std::string &path, const SourceHints &hints)
{
    const bool is_stdin = path == "-";
    FILE *f = is_stdin ? stdin : fopen(path.c_str(), "rb");
    if (!f) fail("could not open '" + path + "'");
    unsigned char probe[PROBE_LEN];
    const size_t got = fread(probe, 1, PROBE_LEN, f);
    auto close = [&] { if (!is_stdin) fclose(f); };
    ImageFormat fmt = ImageFormat::Unknown;
    try {
        if (!hints.force_raw) fmt = probe_image(probe, got);
    } catch (...) { close(); throw; }

    if (fmt != ImageFormat::Unknown) {
        if (!can_decode(fmt)) {
            close();
            fail("'" + path + "' detected as " + to_string(fmt) + " but no decoder is available (missing crate feature or unimplemented).");
        }
        std::vector<unsigned char> data(probe, probe + got);
        unsigned char buf[1 << 16];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
        close();
        return std::make_unique<ImageFrameSource>(std::move(data), fmt);
    }

    if (!hints.force_raw && got >= 10 && !memcmp(probe, "YUV4MPEG2 ", 10)) {
        try {
            return open_y4m_stream(f, std::string((const char *)probe, got), !is_stdin, hints, "'" + path + "'");
        } catch (...) { close(); throw; }
    }

    // compressed video: IVF / Matroska, demuxed here, decoded by an external decoder process (video_input.hpp)
    std::string why_not_video = "UnknownContainer";
    if (!hints.force_raw && !is_stdin && !(hints.width && hints.height)) {
        if (fseek(f, 0, SEEK_SET) == 0) {
            std::unique_ptr<Demuxer> dm;
            try { dm = probe_video(f, why_not_video); } catch (...) { close(); throw; }
            if (dm) return std::make_unique<VideoFrameSource>(std::move(dm), hints); // the demuxer owns the stream now
        }
    }

    if (hints.width && hints.height) { // headerless planar 4:2:0
        FILE *in = f;
        std::vector<unsigned char> prefix;
        if (is_stdin || fseek(f, 0, SEEK_SET) != 0) prefix.assign(probe, probe + got); // a pipe: the probe bytes come first
        const int bits = hints.bits;
        const size_t bps = bits > 8 ? 2 : 1, cw = (hints.width + 1) / 2, ch = (hints.height + 1) / 2;
        const size_t pic = ((size_t)hints.width * hints.height + 2 * cw * ch) * bps;
        const long total = !is_stdin ? file_size_or_zero(f) : 0;
        const ColorCharacteristics cc = ColorCharacteristics::from_codes(hints.cp, hints.mc, hints.tc).or_(color_characteristics_fallback(hints.height));
        auto src = std::make_unique<YuvStreamSource>(in, false, hints.width, hints.height, bits, cc, hints.full_range ? ColorRange::Full : ColorRange::Limited,
                                                     total > 0 ? (size_t)total / pic : 0, "I420" + (bits > 8 ? "p" + std::to_string(bits) : std::string()));
        src->set_prefix(std::move(prefix));
        return src;
    }
    close();
    fail("'" + path + "': not a PNG / PPM / PFM image, a Y4M stream, nor an IVF / Matroska video (" + why_not_video +
         "); for headerless planar YUV give --width/--height");
}

// The YUV4MPEG2 stream whose first bytes are `head` (everything read from `in` so far; the stream header may be longer).
std::unique_ptr<FrameSource> open_y4m_stream(FILE *f, std::string header, bool seekable_file, const SourceHints &hints, const std::string &what)
{
    if (header.size() < 10) {
        char buf[10];
        const size_t need = 10 - header.size(), got = fread(buf, 1, need, f);
        header.append(buf, got);
    }
    if (header.size() < 10 || memcmp(header.data(), "YUV4MPEG2 ", 10)) fail(what + ": not a YUV4MPEG2 stream");
    // the stream header ends at the first '\n'; it may be longer than what was read so far
    size_t nl = header.find('\n');
    while (nl == std::string::npos) {
        const int c = fgetc(f);
        if (c == EOF) fail("Y4M: truncated stream header");
        header.push_back((char)c);
        if (c == '\n') nl = header.size() - 1;
        if (header.size() > 4096) fail("Y4M: stream header too long");
    }
    const std::string rest = header.substr(nl + 1); // bytes of the first FRAME that were already consumed
    std::istringstream ss(header.substr(10, nl - 10));
    uint32_t w = 0, h = 0;
    int bits = 8;
    std::string cs = "420";
    bool full = hints.full_range;
    std::string tok;
    while (ss >> tok) {
        try {
            if (tok[0] == 'W') w = (uint32_t)std::stoul(tok.substr(1));
            else if (tok[0] == 'H') h = (uint32_t)std::stoul(tok.substr(1));
            else if (tok[0] == 'C') cs = tok.substr(1);
            else if (tok == "XCOLORRANGE=FULL") full = true;
            else if (tok == "XCOLORRANGE=LIMITED") full = false;
        } catch (const std::logic_error &) { fail("Y4M: malformed stream header"); }
    }
    if (w == 0 || h == 0) fail("Y4M: missing W/H");
    if (w > MAX_DIM || h > MAX_DIM) check_dims("Y4M", w, h);
    if (cs.rfind("420", 0) != 0) fail("not implemented: Y4M colourspace C" + cs + " (only 4:2:0 reaches the NV12 / P016 surfaces of the reference)");
    const size_t pp = cs.find('p', 3);
    if (pp != std::string::npos && pp + 1 < cs.size() && isdigit((unsigned char)cs[pp + 1])) bits = std::stoi(cs.substr(pp + 1));
    if (bits != 8 && bits != 10 && bits != 12 && bits != 14 && bits != 16) fail("Y4M: unsupported bit depth in C" + cs);
    // more than the stream header may have been read: a seekable file is rewound to the first FRAME, a pipe gets the over-read
    // bytes handed to the source as a prefix that it consumes before touching the stream again
    std::vector<unsigned char> prefix;
    if (!rest.empty() && (!seekable_file || fseek(f, (long)(nl + 1), SEEK_SET) != 0)) prefix.assign(rest.begin(), rest.end());
    const size_t bps = bits > 8 ? 2 : 1, cw = (w + 1) / 2, ch = (h + 1) / 2;
    const size_t pic = ((size_t)w * h + 2 * cw * ch) * bps + 6;
    const long total = seekable_file ? file_size_or_zero(f) : 0;
    const size_t count = total > (long)(nl + 1) ? ((size_t)total - (nl + 1)) / pic : 0;
    const ColorCharacteristics cc = ColorCharacteristics::from_codes(hints.cp, hints.mc, hints.tc).or_(color_characteristics_fallback(h));
    auto src = std::make_unique<YuvStreamSource>(f, true, w, h, bits, cc, full ? ColorRange::Full : ColorRange::Limited, count,
                                                 "I420" + (bits > 8 ? "p" + std::to_string(bits) : std::string()));
    src->set_prefix(std::move(prefix));
    return src;
}

} // namespace tm_host
