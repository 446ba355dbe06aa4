// Microbenchmark (measurement aid, not product): does data that one kernel has just WRITTEN come back faster when another kernel READS
// it while the working set is still smaller than the 256-MB Infinity Cache?  For region sizes S = 16 MB ... 2 GB:
//   write S (plain or non-temporal float4 stores), then read S (sum), timed separately, 5 repetitions, best time;
//   the same read after 2 GB of other data went through the memory system in between ("cold").
// A producer / consumer pair of kernels whose data stays inside the cache would see the first figure; the blur passes of a 64-pair
// launch (10.7 GB between a plane's write and its read) see the second.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ void k_write(f4 *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1.0f, 2.0f, 3.0f, (float)threadIdx.x};
    for (; i < n; i += st) { if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v; }
}
__global__ void k_read(const f4 *__restrict__ a, size_t n, float *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (; i < n; i += st) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.0f;
}

int main()
{
    const size_t MB = 1ull << 20;
    char *buf, *other; float *out;
    CK(hipMalloc(&buf, 2048 * MB)); CK(hipMalloc(&other, 2048 * MB)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int grid = 256 * 16, block = 256;
    for (int i = 0; i < 20; ++i) k_write<false><<<grid, block>>>((f4 *)other, 2048 * MB / 16); // warm the clock
    printf("%8s | %-28s | %-28s | %-20s\n", "MB", "plain store: write, read GB/s", "nt store: write, read GB/s", "cold read GB/s");
    for (size_t S : {16 * MB, 32 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB, 384 * MB, 512 * MB, 1024 * MB, 2048 * MB}) {
        const size_t n = S / 16;
        double best[2][2] = {{0, 0}, {0, 0}}, cold = 0;
        for (int nt = 0; nt < 2; ++nt)
            for (int rep = 0; rep < 5; ++rep) {
                k_write<false><<<grid, block>>>((f4 *)other, 2048 * MB / 16); // push everything else out first
                CK(hipEventRecord(e0));
                if (nt) k_write<true><<<grid, block>>>((f4 *)buf, n); else k_write<false><<<grid, block>>>((f4 *)buf, n);
                CK(hipEventRecord(e1));
                k_read<<<grid, block>>>((const f4 *)buf, n, out);
                CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
                float w, r; CK(hipEventElapsedTime(&w, e0, e1)); CK(hipEventElapsedTime(&r, e1, e2));
                const double gw = S / 1e9 / (w * 1e-3), gr = S / 1e9 / (r * 1e-3);
                if (gw > best[nt][0]) best[nt][0] = gw;
                if (gr > best[nt][1]) best[nt][1] = gr;
            }
        for (int rep = 0; rep < 5; ++rep) {
            k_write<false><<<grid, block>>>((f4 *)buf, n);
            k_write<false><<<grid, block>>>((f4 *)other, 2048 * MB / 16);
            CK(hipEventRecord(e1));
            k_read<<<grid, block>>>((const f4 *)buf, n, out);
            CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
            float r; CK(hipEventElapsedTime(&r, e1, e2));
            const double gr = S / 1e9 / (r * 1e-3);
            if (gr > cold) cold = gr;
        }
        printf("%8zu | %12.0f %12.0f    | %12.0f %12.0f    | %12.0f\n", S / MB, best[0][0], best[0][1], best[1][0], best[1][1], cold);
    }
    return 0;
}
