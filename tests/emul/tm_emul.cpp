// tests/emul/tm_emul.cpp -- TEST INFRASTRUCTURE ONLY: runs the product's kernel source on the CPU,
// one lane at a time, over host buffers laid out exactly like the engine's HBM arenas.
#define TM_EMULATE 1
#include "hip_emul.h"
thread_local uint3_ threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

static unsigned g_accu[3];
static inline unsigned lane_id() { return (threadIdx.x + threadIdx.y * blockDim.x) & 63; }
void tm_emul_wave_barrier();
static bool lockstep_on();
// the device's tm_wave_sum6 is a shuffle tree (a[l] += a[l + off], off = 32 .. 1): the same additions in the same order here, so
// that the f64 partial sums of the emulated kernels carry the bits the GPU's carry
static void tree_sum64(double *v) { for (int off = 32; off > 0; off >>= 1) for (int i = 0; i < off; ++i) v[i] += v[i + off]; }
bool tm_wave_sum6(double (&a)[6])
{
    static double buf[16][6][64]; // per wave of the workgroup
    const unsigned l = lane_id(), wv = ((threadIdx.x + threadIdx.y * blockDim.x) >> 6) & 15;
    if (lockstep_on()) { // the 64 lanes are concurrent fibers / host threads: sum through memory, lane 0 holds the total
        for (int k = 0; k < 6; ++k) buf[wv][k][l] = a[k];
        tm_emul_wave_barrier();
        if (l == 0)
            for (int k = 0; k < 6; ++k) { tree_sum64(buf[wv][k]); a[k] = buf[wv][k][0]; }
        tm_emul_wave_barrier();
        return l == 0;
    }
    // lanes of one wave run back to back (x fastest): the last one adds them up
    for (int k = 0; k < 6; ++k) buf[0][k][l] = a[k];
    if (l == 63) { for (int k = 0; k < 6; ++k) { tree_sum64(buf[0][k]); a[k] = buf[0][k][0]; } return true; }
    return false;
}
bool tm_wave_sum_u32x3(unsigned (&v)[3])
{
    if (lockstep_on()) { // the 64 lanes are concurrent host threads: sum through memory
        static unsigned buf[16][3][64]; // per wave of the workgroup
        const unsigned l = lane_id(), wv = ((threadIdx.x + threadIdx.y * blockDim.x) >> 6) & 15;
        for (int k = 0; k < 3; ++k) buf[wv][k][l] = v[k];
        tm_emul_wave_barrier();
        if (l == 0)
            for (int k = 0; k < 3; ++k) { unsigned t = 0; for (int i = 0; i < 64; ++i) t += buf[wv][k][i]; v[k] = t; }
        tm_emul_wave_barrier();
        return l == 0;
    }
    if (lane_id() == 0) for (int k = 0; k < 3; ++k) g_accu[k] = 0;
    for (int k = 0; k < 3; ++k) g_accu[k] += v[k];
    if (lane_id() == 63) { for (int k = 0; k < 3; ++k) v[k] = g_accu[k]; return true; }
    return false;
}

#include <pthread.h>
#include <thread>
#include <vector>
#include <cstdlib>
#include <cstdio>
static bool g_lockstep = false, g_per_wave = false;
static bool lockstep_on() { return g_lockstep; }

#if defined(__SANITIZE_ADDRESS__) || !defined(__x86_64__)
// ---- lanes as host threads, pthread barriers (AddressSanitizer build: it cannot follow hand-made stack switches) ------------
#define TM_EMUL_FIBERS 0
static pthread_barrier_t g_wave_bar;       // all host threads of the workgroup (__syncthreads)
static pthread_barrier_t g_per_wave_bar[16]; // the 64 host threads of one wavefront (wave-synchronous code)
void tm_emul_wave_barrier()
{
    if (!g_lockstep) return;
    if (g_per_wave) pthread_barrier_wait(&g_per_wave_bar[threadIdx.x >> 6]);
    else pthread_barrier_wait(&g_wave_bar);
}
void tm_emul_syncthreads() { if (g_lockstep) pthread_barrier_wait(&g_wave_bar); }
void tm_emul_yield() { sched_yield(); }
#else
// ---- lanes as cooperative fibers on ONE host thread: a barrier is a yield to the scheduler, a context switch is six pushes.
// (64 .. 320 host threads meeting at pthread barriers millions of times cost minutes of futex traffic per test.)
#define TM_EMUL_FIBERS 1
extern "C" void tm_ctx_switch(void **save_sp, void *new_sp);
asm(".text\n.globl tm_ctx_switch\n.type tm_ctx_switch,@function\ntm_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size tm_ctx_switch,.-tm_ctx_switch\n");
struct TmFiber { void *sp; char *stack; bool done; int wait_bar; unsigned wait_gen, tid; };
struct TmFiberBar { unsigned need, arrived, gen; };
static constexpr size_t kFiberStack = 512 * 1024;
static std::vector<TmFiber> g_fibers;
static TmFiberBar g_fbar[17]; // 0..15: the waves of the workgroup, 16: the whole workgroup
static void *g_sched_sp;
static int g_cur = -1;
static void (*g_fiber_body)(void *);
static void *g_fiber_arg;
static void fiber_barrier(int b)
{
    TmFiberBar &B = g_fbar[b];
    if (++B.arrived == B.need) { B.arrived = 0; ++B.gen; return; } // the last one in releases the others and goes on
    TmFiber &f = g_fibers[g_cur];
    f.wait_bar = b; f.wait_gen = B.gen;
    tm_ctx_switch(&f.sp, g_sched_sp);
}
void tm_emul_wave_barrier()
{
    if (!g_lockstep) return;
    fiber_barrier(g_per_wave ? (int)(threadIdx.x >> 6) : 16);
}
void tm_emul_syncthreads() { if (g_lockstep) fiber_barrier(16); }
// a lane that polls memory written by another wave of its workgroup: let the other fibers run
void tm_emul_yield()
{
    if (!g_lockstep || g_cur < 0) return;
    TmFiber &f = g_fibers[g_cur];
    f.wait_bar = -1;
    tm_ctx_switch(&f.sp, g_sched_sp);
}
static void fiber_entry()
{
    g_fiber_body(g_fiber_arg);
    TmFiber &f = g_fibers[g_cur];
    f.done = true;
    tm_ctx_switch(&f.sp, g_sched_sp);
    abort(); // never resumed
}
// run `n` fibers (thread ids 0..n-1 of the current block) to completion
static void run_fibers(unsigned n, unsigned tid0, void (*body)(void *), void *arg)
{
    if (g_fibers.size() < n) {
        const size_t old = g_fibers.size();
        g_fibers.resize(n);
        for (size_t i = old; i < n; ++i) g_fibers[i].stack = (char *)aligned_alloc(64, kFiberStack);
    }
    g_fiber_body = body; g_fiber_arg = arg;
    for (unsigned i = 0; i < n; ++i) {
        TmFiber &f = g_fibers[i];
        f.done = false; f.wait_bar = -1; f.tid = tid0 + i;
        void **sp = (void **)(f.stack + kFiberStack - 64); // 16-byte aligned; entry sees rsp = 8 mod 16 after the `ret`
        *--sp = nullptr;                 // fake return address of fiber_entry (keeps the ABI alignment)
        *--sp = (void *)fiber_entry;     // `ret` target of the first switch
        for (int k = 0; k < 6; ++k) *--sp = nullptr; // rbp rbx r12 r13 r14 r15
        f.sp = sp;
    }
    unsigned live = n;
    while (live) {
        bool progressed = false;
        for (unsigned i = 0; i < n; ++i) {
            TmFiber &f = g_fibers[i];
            if (f.done) continue;
            if (f.wait_bar >= 0 && g_fbar[f.wait_bar].gen == f.wait_gen) continue; // its barrier is not complete yet
            f.wait_bar = -1;
            g_cur = (int)i;
            threadIdx = {f.tid, 0, 0};
            tm_ctx_switch(&g_sched_sp, f.sp);
            progressed = true;
            if (f.done) --live;
        }
        if (!progressed) { fprintf(stderr, "tm_emul: lanes wait at a barrier that can never complete\n"); abort(); }
    }
    g_cur = -1;
}
#endif

// wave shuffle for the lockstep emulation (64 host threads = the lanes of one wave): exchange through memory
static float g_shfl[16][64]; // per wave of the workgroup: the waves of a lockstep workgroup interleave at their barriers
unsigned tm_shfl_xor_u32(unsigned v, int mask)
{
    static unsigned buf[16][64];
    const unsigned l = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 15;
    tm_emul_wave_barrier();
    buf[wv][l] = v;
    tm_emul_wave_barrier();
    const unsigned r = buf[wv][l ^ (unsigned)mask];
    tm_emul_wave_barrier();
    return r;
}
float tm_shfl_xor(float v, int mask)
{
    const unsigned l = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 15;
    tm_emul_wave_barrier();
    g_shfl[wv][l] = v;
    tm_emul_wave_barrier();
    const float r = g_shfl[wv][l ^ (unsigned)mask];
    tm_emul_wave_barrier();
    return r;
}

#include "../../turbo-metrics_amd/csrc/tm_kernels.h"
#include "../../turbo-metrics_amd/csrc/tm_reference_kernels.h"
#include "../../turbo-metrics_amd/csrc/tm_ssim_kernels.h"

// a whole workgroup of `nthreads` lanes; __syncthreads() is a real barrier
template <typename F> static void launch_wg_lockstep(dim3 grid, unsigned nthreads, F f)
{
#if TM_EMUL_FIBERS
    g_lockstep = true; g_per_wave = true;
    gridDim = grid; blockDim = dim3(nthreads);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = {bx, by, bz};
                for (unsigned i = 0; i < nthreads / 64; ++i) g_fbar[i] = {64u, 0u, 0u};
                g_fbar[16] = {nthreads, 0u, 0u};
                run_fibers(nthreads, 0, [](void *p) { (*(F *)p)(); }, (void *)&f);
            }
    g_lockstep = false; g_per_wave = false;
#else
    pthread_barrier_init(&g_wave_bar, nullptr, nthreads);
    for (unsigned i = 0; i < nthreads / 64; ++i) pthread_barrier_init(&g_per_wave_bar[i], nullptr, 64);
    g_lockstep = true; g_per_wave = true;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t)
        th.emplace_back([=] {
            gridDim = grid; blockDim = dim3(nthreads);
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        blockIdx = {bx, by, bz}; threadIdx = {t, 0, 0};
                        pthread_barrier_wait(&g_wave_bar);
                        f();
                        pthread_barrier_wait(&g_wave_bar);
                    }
        });
    for (auto &x : th) x.join();
    g_lockstep = false; g_per_wave = false;
    pthread_barrier_destroy(&g_wave_bar);
    for (unsigned i = 0; i < nthreads / 64; ++i) pthread_barrier_destroy(&g_per_wave_bar[i]);
#endif
}

// one workgroup == `nwaves` wavefronts that never talk to each other: one after the other, 64 lanes in lockstep-by-barrier
template <typename F> static void launch_wave_lockstep(dim3 grid, F f, unsigned nwaves = 1)
{
#if TM_EMUL_FIBERS
    g_lockstep = true;
    gridDim = grid; blockDim = dim3(64 * nwaves);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx)
                for (unsigned wv = 0; wv < nwaves; ++wv) {
                    blockIdx = {bx, by, bz};
                    g_fbar[16] = {64u, 0u, 0u};
                    run_fibers(64, wv * 64, [](void *p) { (*(F *)p)(); }, (void *)&f);
                }
    g_lockstep = false;
#else
    pthread_barrier_init(&g_wave_bar, nullptr, 64);
    g_lockstep = true;
    std::vector<std::thread> th;
    for (unsigned lane = 0; lane < 64; ++lane)
        th.emplace_back([=] {
            gridDim = grid; blockDim = dim3(64 * nwaves);
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx)
                        for (unsigned wv = 0; wv < nwaves; ++wv) { // waves of a workgroup that never talk: one after the other
                            blockIdx = {bx, by, bz}; threadIdx = {wv * 64 + lane, 0, 0};
                            pthread_barrier_wait(&g_wave_bar);
                            f();
                            pthread_barrier_wait(&g_wave_bar);
                        }
        });
    for (auto &t : th) t.join();
    g_lockstep = false;
    pthread_barrier_destroy(&g_wave_bar);
#endif
}

template <typename F> static void launch(dim3 grid, dim3 block, F f)
{
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = {bx, by, bz};
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx) { threadIdx = {tx, ty, tz}; f(); }
            }
}

extern "C" {

void emul_geom(int w, int h, TmGeom *g) { tm_make_geom(g, w, h); }
size_t emul_geom_size(void) { return sizeof(TmGeom); }

// counts (in elements) the caller must allocate: LIN, XYB (floats), XYBT, V, PART (doubles), SUMS, SSE
void emul_sizes(int w, int h, int n, unsigned long long out[7])
{
    TmGeom g; tm_make_geom(&g, w, h);
    out[0] = (unsigned long long)n * 2 * g.pyr; out[1] = out[0];
    out[2] = (unsigned long long)n * 2 * g.pyr_t; out[3] = (unsigned long long)n * 5 * g.pyr_t;
    out[4] = (unsigned long long)n * 3 * g.hblk[TM_SCALES] * 6; out[5] = (unsigned long long)n * 108; out[6] = (unsigned long long)n * TM_SSE_BINS * 3;
}

static int g_ingest_rows = 4; // quad rows per wave of k_ingest_rows (even)
void emul_set_ingest_rows(int r) { g_ingest_rows = r < 2 ? 2 : (r & ~1); }

// variant: 0 = the default pipeline, 1 = the reference pipeline (TM_VARIANT_REFERENCE), 0x100 = default with the wide-frame row pass,
// 0x200 = default with the tile ingest kernel for the 4:2:0 kinds too (TM_VARIANT_TILE_INGEST), 0x400 = the three-wave row pass
// of small batches (TM_VARIANT_SPLIT_ROWS)
void emul_pipeline(int w, int h, int n, const TmFrameDesc *desc, const float *lut, const float *coef, const double *tab, int want_sse,
                   float *LIN, float *XYB, float *XYBT, float *V, double *PART, double *SUMS, unsigned long long *SSE,
                   int variant, const double *weights, int full_sums, unsigned char *QU8, unsigned long long qplane, int qpitch)
{
    const bool reference = (variant & 1) != 0, wide_rows = (variant & 0x100) != 0;
    TmGeom g; tm_make_geom(&g, w, h);
    const bool fused_edge = (variant & 0x4000) != 0 && !reference; // TM_VARIANT_FUSED_EDGE: the EDGE jobs through k_blur_edge_fused
    TmJobs jobs; tm_make_jobs(&jobs, &g, weights, full_sums, fused_edge ? 1 : 0);
    const int qw = (w + 1) / 2, qh = (h + 1) / 2;
    if (reference) {
        launch(dim3((qw + 63) / 64, (qh + 3) / 4, n), dim3(64, 4, 1), [&] { tmk::k_ingest(g, desc, lut, coef, tab, LIN, SSE, want_sse); });
        for (int s = 1; s < TM_SCALES; ++s)
            launch(dim3((g.s[s].w + 63) / 64, g.s[s].h, n * 6), dim3(64), [&] { tmk::k_downscale(g, s, LIN); });
        for (int s = 0; s < TM_SCALES; ++s)
            launch(dim3((g.s[s].w + 63) / 64, g.s[s].h, n * 2), dim3(64), [&] { tmk::k_xyb(g, s, LIN, XYB); });
        launch(dim3(g.vblk[TM_SCALES], 3, n), dim3(64), [&] { tmk::k_blur_v(g, XYB, XYBT, V); });
        launch(dim3(jobs.hstart[TM_MAX_JOBS], 1, n), dim3(64), [&] { tmk::k_blur_h_jobs(g, jobs, XYBT, V, PART); });
    } else {
        std::vector<float> lin2((size_t)n * 2 * 3 * g.s[2].plane, 0.0f);
        int kind = desc[0].kind;
        for (int i = 1; i < 2 * n; ++i) if (desc[i].kind != kind) kind = -1;
        const bool yuv = kind == TM_KIND_NV12 || kind == TM_KIND_P016 || kind == TM_KIND_I420_8 || kind == TM_KIND_I420_16 || kind == TM_KIND_I420_P10;
        bool folded = false;
        if (yuv && !(variant & 0x200)) { // the engine's choice for the 4:2:0 kinds: the side-packed row-walking kernel
            const int rpw = g_ingest_rows;
            folded = !(variant & 0x2000) && (rpw == 4 || rpw == 8); // TM_VARIANT_UPPER_KERNEL / other row counts: levels 2..5 by k_ingest_upper_rd
            if (folded) launch_wg_lockstep(dim3((qw + 63) / 64, (qh + 4 * rpw - 1) / (4 * rpw), n), 256, [&] {
                switch (kind) {
                case TM_KIND_NV12: tmk::k_ingest_rows<TM_KIND_NV12, true, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_P016: tmk::k_ingest_rows<TM_KIND_P016, true, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_I420_8: tmk::k_ingest_rows<TM_KIND_I420_8, true, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_I420_P10: tmk::k_ingest_rows<TM_KIND_I420_P10, true, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                default: tmk::k_ingest_rows<TM_KIND_I420_16, true, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                } });
            else
            launch_wg_lockstep(dim3((qw + 63) / 64, (qh + 4 * rpw - 1) / (4 * rpw), n), 256, [&] {
                switch (kind) {
                case TM_KIND_NV12: tmk::k_ingest_rows<TM_KIND_NV12, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_P016: tmk::k_ingest_rows<TM_KIND_P016, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_I420_8: tmk::k_ingest_rows<TM_KIND_I420_8, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                case TM_KIND_I420_P10: tmk::k_ingest_rows<TM_KIND_I420_P10, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                default: tmk::k_ingest_rows<TM_KIND_I420_16, true>(tmk::tm_ingest_geom(g), desc, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch, rpw); break;
                } });
        } else
        launch_wave_lockstep(dim3((w + 31) / 32, (h + 7) / 8, n), [&] {
            switch (kind) {
            case TM_KIND_NV12: tmk::k_ingest_wave<TM_KIND_NV12>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            case TM_KIND_P016: tmk::k_ingest_wave<TM_KIND_P016>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            case TM_KIND_I420_8: tmk::k_ingest_wave<TM_KIND_I420_8>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            case TM_KIND_I420_16: tmk::k_ingest_wave<TM_KIND_I420_16>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            case TM_KIND_I420_P10: tmk::k_ingest_wave<TM_KIND_I420_P10>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            default: tmk::k_ingest_wave<-1>(g, desc, lut, coef, tab, XYB, lin2.data(), SSE, want_sse, QU8, qplane, qpitch); break;
            } });
        if (!folded) launch_wg_lockstep(dim3((g.s[2].w + 31) / 32, (g.s[2].h + 31) / 32, n), 256, [&] { tmk::k_ingest_upper_rd(g, lin2.data(), XYB); });
        const int vb = jobs.vstart[jobs.nfull], hb = jobs.hstart[jobs.nfull]; // the two passes run jobs [0, nfull)
        // 0x10000: the column pass with every role-wave as a workgroup of its own (launches of a pair or two)
        if (variant & 0x10000) launch_wave_lockstep(dim3(n, vb, 5), [&] { tmk::k_blur_v_jobs<32, 16, 0, true>(g, jobs, XYB, V); });
        else launch_wave_lockstep(dim3(n, vb, 1), [&] { tmk::k_blur_v_jobs<32, 16>(g, jobs, XYB, V); }, 5);
        // 0x400: the eight-wave row pass of small launches (TM_VARIANT_SPLIT_ROWS)
        if (variant & 0x400) launch_wg_lockstep(dim3(n, hb, 1), 64 * TM_SPLIT_WAVES, [&] { tmk::k_blur_h_jobs_split(g, jobs, XYB, V, PART); });
        else if (wide_rows) launch_wave_lockstep(dim3(n, hb, 1), [&] { tmk::k_blur_h_jobs_x<16, 8, 16, 8>(g, jobs, XYB, V, PART); });
        else launch_wave_lockstep(dim3(n, hb, 1), [&] { tmk::k_blur_h_jobs_x<16, 8, 32, 16>(g, jobs, XYB, V, PART); });
        if (jobs.n > jobs.nfull) { // the engine's buffers of the fused EDGE kernel, sized the same way
            const int ne = jobs.n - jobs.nfull;
            int tiles = 0, bands = 0;
            for (int k = jobs.nfull; k < jobs.n; ++k) { tiles = std::max(tiles, (g.s[jobs.scale[k]].w + 31) / 32); bands = std::max(bands, (g.s[jobs.scale[k]].h + 31) / 32); }
            std::vector<unsigned long long> hs((size_t)n * ne * 2 * tiles * 384, 0ull);
            std::vector<double> erows((size_t)n * ne * bands * 128, 0.0);
            unsigned epoch[2] = {1u, 0u}; int status = 0; // launch epoch, ticket counter
            for (int rep = 0; rep < 2; ++rep) { // twice: the second launch finds the first one's words (tags of another epoch) in HS
                tmk::TmEdgeArgs ea;
                tmk::tm_make_edge_args(&ea, &g, &jobs, tiles, bands);
                if (variant & 0x8000) { // four adjacent bands of one plane per workgroup (what the engine launches); second launch: a third of the workgroups share the tickets
                    const int groups = (bands + 3) / 4, total = n * ne * groups;
                    launch_wg_lockstep(dim3(rep == 0 ? total : std::max(1, total / 3), 1, 1), 256, [&] { tmk::k_blur_edge_fused<4, true>(ea, n * ne, groups, (unsigned)total, XYB, hs.data(), epoch, epoch + 1, erows.data(), &status); });
                } else
                launch_wave_lockstep(dim3(rep == 0 ? n * ne * bands : std::max(1, n * ne * bands / 3), 1, 1), [&] { tmk::k_blur_edge_fused<1, false>(ea, n * ne, n * ne, (unsigned)(n * ne * bands), XYB, hs.data(), epoch, epoch + 1, erows.data(), &status); });
                launch(dim3(n * ne), dim3(64), [&] { tmk::k_finish_edge(ea, erows.data(), PART, epoch); });
            }
            if (status != 0 || epoch[0] != 3 || epoch[1] != 0) { fprintf(stderr, "tm_emul: k_blur_edge_fused status %d epoch %u\n", status, epoch[0]); abort(); }
        }
    }
    launch(dim3(n), dim3(128), [&] { tmk::k_finish_jobs(jobs, PART, SUMS); });
    if (!reference) { // test convenience: turn the interleaved pyramid back into two plain ones for the plane checks
        std::vector<float> tmp((size_t)2 * g.pyr);
        for (int sl = 0; sl < n; ++sl) {
            float *base = XYB + (size_t)sl * 2 * g.pyr;
            for (size_t i = 0; i < g.pyr; ++i) { tmp[i] = base[2 * i]; tmp[g.pyr + i] = base[2 * i + 1]; }
            for (size_t i = 0; i < 2 * g.pyr; ++i) base[i] = tmp[i];
        }
    }
}

size_t emul_ssim_geom_size(void) { return sizeof(TmSsimGeom); }
void emul_ssim_geom(int w, int h, const float *g, TmSsimGeom *out) { tm_make_ssim_geom(out, w, h, g); }

// SSIM / MS-SSIM stage on the planar u8 planes the ingest kernel left in QU8: pyramid, statistics, finisher.
// PYR: n*2*3*sg.pyr u16, PART: n*3*item_off[5]*2 doubles, SUMS: n*30 doubles
void emul_ssim(int w, int h, int n, const float *g, const unsigned char *QU8, unsigned short *PYR, double *PART, double *SUMS, unsigned need_l)
{
    TmSsimGeom sg; tm_make_ssim_geom(&sg, w, h, g);
    if (sg.w[1] > 0 && sg.h[1] > 0)
        launch_wave_lockstep(dim3((w + 127) / 128, (h + 31) / 32, n * 6), [&] { tmk::k_ssim_pyramid(sg, QU8, PYR); });
    int nscales = 0;
    for (int s = 0; s < TM_SSIM_SCALES; ++s) if (sg.strips_x[s] > 0 && sg.segs_y[s] > 0) nscales = s + 1;
    if (nscales > 0) launch_wave_lockstep(dim3(n * 3, sg.item_off[nscales], 1), [&] { tmk::k_ssim_stream(sg, nscales, need_l, QU8, PYR, PART); });
    launch(dim3(n, 30, 1), dim3(64), [&] { tmk::k_ssim_finish(sg, nscales, PART, SUMS); });
}
}
