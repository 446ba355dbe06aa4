// Microbenchmark: 2-D tiled single-visit writes. A wave writes a tile of 64 rows x RUN bytes once (like the ingest
// kernel's stores), tiles cover the plane; consecutive blockIdx.x -> consecutive tiles along the row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int RUN, int ROWS>
__global__ void __launch_bounds__(64) k_tile2d(char* __restrict__ base, size_t pitch, int tiles_x, int delay) {
    constexpr int LPR = RUN / 16, RPI = 64 / LPR;
    const int lane = threadIdx.x, rl = lane / LPR, q = lane % LPR;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    float4 v = make_float4(lane, 1, 2, 3);
    for (int d = 0; d < delay; ++d) v.x = __builtin_fmaf(v.x, 1.0001f, 0.5f);
#pragma unroll
    for (int i = 0; i < ROWS / RPI; ++i) {
        const size_t row = (size_t)by * ROWS + i * RPI + rl;
        *(float4*)(base + row * pitch + (size_t)bx * RUN + q * 16) = v;
    }
}
template <int RUN, int ROWS> void run(char* buf, size_t bytes, size_t pitch, int delay) {
    size_t rows = bytes / pitch / ROWS * ROWS; int tiles_x = (int)(pitch / RUN); size_t nblk = rows / ROWS * tiles_x;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_tile2d<RUN, ROWS><<<nblk, 64>>>(buf, pitch, tiles_x, delay);
    CK(hipEventRecord(a));
    for (int it = 0; it < 3; ++it) k_tile2d<RUN, ROWS><<<nblk, 64>>>(buf, pitch, tiles_x, delay);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("tile %4d B x %3d rows, pitch %6zu, delay %4d: %.1f GB/s\n", RUN, ROWS, pitch, delay, (double)rows * tiles_x * RUN * 3 / ms / 1e6);
}
int main() {
    const size_t bytes = 4ull << 30; char* b; CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes));
    for (int delay : {0, 500}) for (size_t pitch : {(size_t)7680, (size_t)4352}) {
        run<64, 32>(b, bytes, pitch / 64 * 64, delay); run<128, 32>(b, bytes, pitch / 128 * 128, delay); run<256, 32>(b, bytes, pitch / 256 * 256, delay);
        run<512, 32>(b, bytes, pitch / 512 * 512, delay); run<128, 64>(b, bytes, pitch / 128 * 128, delay); run<256, 64>(b, bytes, pitch / 256 * 256, delay);
    }
    return 0;
}
