// Microbenchmark (measurement aid, not product): what is the plain-store ceiling of this MI355X, and does it depend on the
// shape of the store?  VERDICT r05 #1: tools/microbench/wrpattern.hip measured 5.2 TB/s for a float4 fill of 4 GB and 5.4 TB/s
// for 128-B runs, while MI355X_MICROARCH.md quotes 6.0-6.2 TB/s for "one dword per lane, 256 B per wave-instruction, random
// 2,304-B rows of a 75 MB or 302 MB table, 8 waves per CU".  This program runs the guide's shape next to the fill, over table
// sizes from 75 MB (inside the 256-MiB Infinity Cache) to 4.8 GB (far outside), plain and non-temporal, and the column pass'
// flush shape (a float4 per lane, runs of 128 B or 256 B at a transposed pitch) over the same sizes.
//   hipcc --offload-arch=gfx950 -O3 -o wr_ceiling wr_ceiling.hip && ./wr_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// the guide's shape: a wave writes whole rows of ROWB bytes (ROWB / 256 wave-instructions of one dword per lane), rows drawn at
// random from a table of `rows` rows; `per_wave` rows per wave
template <bool NT>
__global__ void __launch_bounds__(64) k_rows_dword(float *__restrict__ tab, unsigned rows, unsigned per_wave, unsigned rowb)
{
    const unsigned lane = threadIdx.x, wave = blockIdx.x;
    const float v = (float)lane;
    for (unsigned i = 0; i < per_wave; ++i) {
        const unsigned r = mix(wave * per_wave + i) % rows;
        float *row = tab + (size_t)r * (rowb / 4);
        for (unsigned k = 0; k < rowb / 256; ++k) {
            if (NT) __builtin_nontemporal_store(v, row + k * 64 + lane);
            else row[k * 64 + lane] = v;
        }
    }
}

// sequential fill, grid-stride: DW = 1 (one dword per lane, 256 B per wave-instruction) or 4 (float4, 1 KB per wave-instruction)
template <int DW, bool NT>
__global__ void __launch_bounds__(256) k_fill(float *__restrict__ b, size_t n_dw)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * DW;
    const size_t st = (size_t)gridDim.x * blockDim.x * DW;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (; i + DW <= n_dw; i += st) {
        if (DW == 4) { f4 v = {1, 2, 3, 4}; if (NT) __builtin_nontemporal_store(v, (f4 *)(b + i)); else *(f4 *)(b + i) = v; }
        else { if (NT) __builtin_nontemporal_store(1.0f, b + i); else b[i] = 1.0f; }
    }
}

// the column pass' flush: a wave owns 64 consecutive transposed rows (= image columns) of `pitch` bytes and advances along them;
// each store instruction is a float4 per lane, RUN / 16 lanes per row, 64 / (RUN / 16) rows per instruction
template <int RUN, bool NT>
__global__ void __launch_bounds__(64) k_flush(char *__restrict__ base, size_t pitch, int nflush)
{
    constexpr int LPR = RUN / 16, RPI = 64 / LPR;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x, rl = lane / LPR, q = lane % LPR;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const f4 v = {(float)lane, 1, 2, 3};
    for (int f = 0; f < nflush; ++f)
#pragma unroll
        for (int i = 0; i < 64 / RPI; ++i) {
            f4 *p = (f4 *)(base + (row0 + i * RPI + rl) * pitch + (size_t)f * RUN + q * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
}

template <typename F> double time_ms(F launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms / reps;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.gcnArchName, cus);
    const size_t cap = (size_t)4800 << 20;
    char *buf; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 0, cap));
    // warm the clock
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_fill<4, false>), dim3(cus * 32), dim3(256), 0, 0, (float *)buf, cap / 4);
    CK(hipDeviceSynchronize());
    const size_t sizes[] = {(size_t)75 << 20, (size_t)302 << 20, (size_t)1200 << 20, (size_t)4800 << 20};
    printf("# shape, table MB, plain GB/s, non-temporal GB/s\n");
    for (size_t T : sizes) {
        const unsigned rowb = 2304, rows = (unsigned)(T / rowb);
        for (int wpc : {8, 16}) {
            // every launch writes ~4.8 GB so that small tables are overwritten many times within one timing
            const unsigned waves = (unsigned)cus * wpc;
            const unsigned per_wave = (unsigned)((cap / rowb) / waves);
            const double bytes = (double)waves * per_wave * rowb;
            const double p = time_ms([&] { hipLaunchKernelGGL((k_rows_dword<false>), dim3(waves), dim3(64), 0, 0, (float *)buf, rows, per_wave, rowb); }, 3);
            const double n = time_ms([&] { hipLaunchKernelGGL((k_rows_dword<true>), dim3(waves), dim3(64), 0, 0, (float *)buf, rows, per_wave, rowb); }, 3);
            printf("random 2304-B rows, dword per lane, %2d waves/CU, %5zu, %7.0f, %7.0f\n", wpc, T >> 20, bytes / p / 1e6, bytes / n / 1e6);
        }
    }
    for (size_t T : sizes) {
        const int reps = (int)(cap / T) > 3 ? (int)(cap / T) : 3;
        const double a = time_ms([&] { hipLaunchKernelGGL((k_fill<1, false>), dim3(cus * 8), dim3(256), 0, 0, (float *)buf, T / 4); }, reps);
        const double b = time_ms([&] { hipLaunchKernelGGL((k_fill<1, true>), dim3(cus * 8), dim3(256), 0, 0, (float *)buf, T / 4); }, reps);
        printf("sequential fill, dword per lane (256 B / instr), 8 waves/CU, %5zu, %7.0f, %7.0f\n", T >> 20, T / a / 1e6, T / b / 1e6);
        const double c = time_ms([&] { hipLaunchKernelGGL((k_fill<4, false>), dim3(cus * 32), dim3(256), 0, 0, (float *)buf, T / 4); }, reps);
        const double d = time_ms([&] { hipLaunchKernelGGL((k_fill<4, true>), dim3(cus * 32), dim3(256), 0, 0, (float *)buf, T / 4); }, reps);
        printf("sequential fill, float4 per lane (1 KB / instr), 32 waves/CU, %5zu, %7.0f, %7.0f\n", T >> 20, T / c / 1e6, T / d / 1e6);
    }
    // the flush shape over an arena of T bytes: pitch 4352 B (a 1080p column of the transposed planes: 1088 floats)
    for (size_t T : {sizes[2], sizes[3]}) { // (one wave per 64 rows: the small tables would not fill the chip)
        const size_t pitch = 4352;
        const size_t rows = T / pitch / 64 * 64;
        const int reps = (int)(cap / T) > 3 ? (int)(cap / T) : 3;
        auto by = [&](size_t run) { return (double)rows * (double)(pitch / run * run); };
#define FL(RUN, NT) time_ms([&] { hipLaunchKernelGGL((k_flush<RUN, NT>), dim3((unsigned)(rows / 64)), dim3(64), 0, 0, buf, pitch, (int)(pitch / RUN)); }, reps)
        const double a = FL(128, false), b = FL(128, true), c = FL(256, false), d = FL(256, true), e = FL(512, false), f = FL(512, true);
        printf("flush 128-B runs (float4 x 8 lanes x 8 rows), %5zu, %7.0f, %7.0f\n", T >> 20, by(128) / a / 1e6, by(128) / b / 1e6);
        printf("flush 256-B runs (float4 x 16 lanes x 4 rows), %5zu, %7.0f, %7.0f\n", T >> 20, by(256) / c / 1e6, by(256) / d / 1e6);
        printf("flush 512-B runs (float4 x 32 lanes x 2 rows), %5zu, %7.0f, %7.0f\n", T >> 20, by(512) / e / 1e6, by(512) / f / 1e6);
    }
    return 0;
}
