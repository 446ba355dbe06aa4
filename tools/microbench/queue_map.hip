// queue_map.hip -- which streams of a process share a hardware queue?  The HIP runtime binds a stream to one of GPU_MAX_HW_QUEUES (4)
// hardware queues when the stream is created, and kernels of two streams on one queue run one after the other.  The probe creates streams
// in the order an engine does (engine, side, upload; then further engines), optionally touching the null stream in between like
// tm_engine_create's hipMemset / hipMemcpy did, and times a pair of 2-ms one-workgroup spin kernels on every pair of streams: ~2 ms =
// different queues, ~4 ms = one queue.
// build: hipcc --offload-arch=gfx950 -O2 -o queue_map queue_map.hip      usage: queue_map [touch_null 0|1] [streams]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 1000) *sink = 1;
}

int main(int argc, char **argv)
{
    const int touch_null = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<hipStream_t> s((size_t)n);
    int *d = nullptr;
    hipMalloc((void **)&d, 256);
    for (int i = 0; i < n; ++i) {
        hipStreamCreateWithFlags(&s[(size_t)i], hipStreamNonBlocking);
        if (i == 2 && touch_null) { hipMemset(d, 0, 256); hipStreamSynchronize(nullptr); } // after engine, side, upload: the first engine's allocations
    }
    const long long cycles = 200000; // wall_clock64 runs at 100 MHz: 2 ms
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[(size_t)i], 1000, d); hipStreamSynchronize(s[(size_t)i]); }
    printf("# touch_null %d, %d streams; pair time in ms (2 = beside each other, 4 = one queue)\n     ", touch_null, n);
    for (int j = 0; j < n; ++j) printf(" s%-4d", j);
    printf(" null\n");
    for (int i = 0; i < n; ++i) {
        printf("s%-4d", i);
        for (int j = 0; j <= n; ++j) {
            if (j <= i && j < n) { printf("   .  "); continue; }
            hipStream_t b = j < n ? s[(size_t)j] : nullptr;
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[(size_t)i], cycles, d);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, cycles, d);
            hipStreamSynchronize(s[(size_t)i]);
            hipStreamSynchronize(b);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf(" %5.1f", ms);
        }
        printf("\n");
    }
    return 0;
}
