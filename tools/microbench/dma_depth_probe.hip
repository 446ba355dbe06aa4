// dma_depth_probe: host-to-device copies of 3.1-MB pictures out of a page-locked ring the way the CLI issues them -- a bounded number in
// flight, the next one submitted when the oldest is done -- on one stream, and alternating between two streams; with and without a kernel
// streaming through HBM beside them.  usage: dma_depth_probe [picture bytes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_stream(float4 *p, size_t n, int reps)
{
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.0f; p[i] = v; }
}
int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t pic = argc > 1 ? (size_t)atoll(argv[1]) : 3110400, N = 1024, R = 64;
    (void)hipSetDevice(0);
    unsigned char *ring, *dev; float4 *big;
    if (hipHostMalloc(&ring, R * pic, 0) != hipSuccess || hipMalloc(&dev, R * pic) != hipSuccess || hipMalloc(&big, (size_t)2 << 30) != hipSuccess) return 1;
    for (size_t i = 0; i < R * pic; i += 4096) ring[i] = (unsigned char)i;
    hipStream_t st[2], ks; hipEvent_t ev[64];
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&ks, hipStreamNonBlocking);
    for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int busy = 0; busy < 2; ++busy) {
        printf(busy ? "== with a kernel streaming through HBM on another stream\n" : "== copies alone\n");
        for (int nst = 1; nst <= 2; ++nst)
            for (int depth : {1, 2, 4, 8, 16, 32, 0}) {
                if (busy) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, ks, big, ((size_t)2 << 30) / 16, 400);
                double t0 = now();
                for (size_t p = 0; p < N; ++p) {
                    if (depth && p >= (size_t)depth) (void)hipEventSynchronize(ev[(p - depth) % 64]);
                    hipStream_t s = st[nst == 2 ? p & 1 : 0];
                    (void)hipMemcpyAsync(dev + (p % R) * pic, ring + (p % R) * pic, pic, hipMemcpyHostToDevice, s);
                    if (depth) (void)hipEventRecord(ev[p % 64], s);
                }
                (void)hipStreamSynchronize(st[0]); (void)hipStreamSynchronize(st[1]);
                double t1 = now();
                printf("  %d stream(s), %2d copies in flight: %.1f GB/s (%.0f us per copy)\n", nst, depth, N * pic / (t1 - t0) / 1e9, (t1 - t0) / N * 1e6);
                if (busy) (void)hipStreamSynchronize(ks);
            }
    }
    return 0;
}
