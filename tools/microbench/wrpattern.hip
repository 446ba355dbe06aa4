// Microbenchmark (measurement aid, not product): how fast can MI355X absorb "transposed tile" writes?
// Each wave repeatedly writes NROWS runs of RUN bytes (one float4 per lane, RUN/16 lanes per run), the runs
// being `pitch` bytes apart, then advances along the runs -- the store pattern of the column pass's flush.
// Also: plain float4 copy and plain float4 write for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void k_fill(float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = make_float4(1, 2, 3, 4);
}
// plane: rows x pitch bytes. A wave owns 64 consecutive rows... each flush writes RUN bytes into each of its 64 rows
// at column offset y0; `planes` planes (separate regions) per flush; steps = pitch/RUN flushes.
template <int RUN>
__global__ void __launch_bounds__(64) k_tiles(char* __restrict__ base, size_t plane_bytes, int planes, size_t pitch, int nflush, int delay) {
    constexpr int LPR = RUN / 16;        // lanes per run
    constexpr int RPI = 64 / LPR;        // runs (rows) per store instruction
    const int lane = threadIdx.x;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int rl = lane / LPR, q = lane % LPR;
    float4 v = make_float4(lane, 1, 2, 3);
    for (int f = 0; f < nflush; ++f) {
        for (int p = 0; p < planes; ++p) {
            char* pb = base + (size_t)p * plane_bytes;
#pragma unroll
            for (int i = 0; i < 64 / RPI; ++i) {
                const size_t row = row0 + i * RPI + rl;
                *(float4*)(pb + row * pitch + (size_t)f * RUN + q * 16) = v;
            }
        }
        // emulate compute between flushes
        for (int d = 0; d < delay; ++d) { v.x = __builtin_fmaf(v.x, 1.0001f, 0.5f); }
        __builtin_amdgcn_s_sleep(0);
    }
    if (v.x == 12345.678f) base[0] = 1;
}

template <int RUN> float run_tiles(char* buf, size_t plane_bytes, int planes, size_t pitch, int rows, int delay) {
    int nflush = (int)(pitch / RUN);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_tiles<RUN><<<rows / 64, 64>>>(buf, plane_bytes, planes, pitch, nflush, delay);
    CK(hipEventRecord(a));
    for (int it = 0; it < 3; ++it) k_tiles<RUN><<<rows / 64, 64>>>(buf, plane_bytes, planes, pitch, nflush, delay);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main() {
    const size_t GB = 1ull << 30;
    char *a, *b;
    const size_t bytes = 4 * GB;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    k_copy<<<2048 * 4, 256>>>((float4*)a, (float4*)b, bytes / 16);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) k_copy<<<2048 * 4, 256>>>((float4*)a, (float4*)b, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("copy   : %.1f GB/s (read+write)\n", 2.0 * bytes * 3 / ms / 1e6);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) k_fill<<<2048 * 4, 256>>>((float4*)b, bytes / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fill   : %.1f GB/s (write)\n", 1.0 * bytes * 3 / ms / 1e6);
    // tiles: 7 planes, pitch 4352 B (1080p transposed), rows = as many as fit in 4 GB
    for (int delay : {0, 2000}) {
        for (size_t pitch : {(size_t)4352, (size_t)8704}) {
            const int planes = 7;
            size_t rows = bytes / planes / pitch / 64 * 64;
            if (rows > 64 * 16384) rows = 64 * 16384;
            size_t plane_bytes = rows * pitch;
            double total = (double)planes * rows * (pitch / 1024 * 1024);
            float t;
            t = run_tiles<16>(b, plane_bytes, planes, pitch, (int)rows, delay);   printf("tiles delay %4d pitch %zu run   16 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 16 * 16) / t / 1e6);
            t = run_tiles<32>(b, plane_bytes, planes, pitch, (int)rows, delay);   printf("tiles delay %4d pitch %zu run   32 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 32 * 32) / t / 1e6);
            t = run_tiles<64>(b, plane_bytes, planes, pitch, (int)rows, delay);   printf("tiles delay %4d pitch %zu run   64 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 64 * 64) / t / 1e6);
            t = run_tiles<128>(b, plane_bytes, planes, pitch, (int)rows, delay);  printf("tiles delay %4d pitch %zu run  128 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 128 * 128) / t / 1e6);
            t = run_tiles<256>(b, plane_bytes, planes, pitch, (int)rows, delay);  printf("tiles delay %4d pitch %zu run  256 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 256 * 256) / t / 1e6);
            t = run_tiles<512>(b, plane_bytes, planes, pitch, (int)rows, delay);  printf("tiles delay %4d pitch %zu run  512 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 512 * 512) / t / 1e6);
            t = run_tiles<1024>(b, plane_bytes, planes, pitch, (int)rows, delay); printf("tiles delay %4d pitch %zu run 1024 B: %.1f GB/s\n", delay, pitch, (double)planes * rows * (pitch / 1024 * 1024) / t / 1e6);
            (void)total;
        }
    }
    return 0;
}
