// mmap_dma_probe: can the CLI skip its CPU copy (file -> page-locked ring) by page-locking the MAPPING of the clip itself and letting
// the copy engines read the page cache?  Writes a file of S MB in <dir>, maps it read-only, registers it with hipHostRegister (whole, and
// in windows of 64 MB), and times host-to-device copies of 3-MB "pictures" from (a) the registered mapping, (b) a hipHostMalloc ring fed
// by pread (what the CLI does today, one thread), (c) the unregistered mapping.  build: hipcc -O2 --offload-arch=gfx950 -o mmap_dma_probe mmap_dma_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t mb = argc > 2 ? (size_t)atoll(argv[2]) : 2048, S = mb << 20, pic = 3110400 + 6, npic = S / pic;
    const std::string path = dir + "/mmap_dma_probe.bin";
    {
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<unsigned char> buf(8 << 20);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = (unsigned char)(i * 2654435761u >> 13);
        for (size_t done = 0; done < S; done += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) return 2;
        close(fd);
    }
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int mode = argc > 3 ? atoi(argv[3]) : 0; // 0: registered mapping (read-only flag), 1: + plain flag, 2: + the unregistered mapping
    CK(hipSetDevice(0));
    unsigned char *dev; const size_t dpic = 2048 * 1080 + 1024 * 1080; CK(hipMalloc(&dev, 64 * dpic));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int fd = open(path.c_str(), O_RDONLY);
    for (int populate = 0; populate < 2; ++populate) {
        double t0 = now();
        unsigned char *m = (unsigned char *)mmap(nullptr, S, PROT_READ, MAP_SHARED | (populate ? MAP_POPULATE : 0), fd, 0);
        double t1 = now();
        printf("mmap%s of %zu MB: %.1f ms\n", populate ? " MAP_POPULATE" : "", mb, (t1 - t0) * 1e3);
        // (c) unregistered mapping
        if (!populate && mode >= 2) {
            t0 = now();
            for (size_t p = 0; p < std::min<size_t>(npic, 128); ++p) CK(hipMemcpyAsync(dev + (p % 64) * pic, m + p * pic, pic, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st)); t1 = now();
            printf("  copies from the UNREGISTERED mapping: %.2f GB/s\n", std::min<size_t>(npic, 128) * pic / (t1 - t0) / 1e9);
        }
        for (unsigned flags : {(unsigned)hipHostRegisterReadOnly, 0u}) {
            if (flags == 0 && mode < 1) continue;
            {   // one window, one picture, first
                hipError_t e1 = hipHostRegister(m, 64 << 20, flags);
                printf("  one 64-MB window (flags %u): %s\n", flags, hipGetErrorString(e1));
                if (e1 != hipSuccess) { (void)hipGetLastError(); continue; }
                CK(hipMemcpyAsync(dev, m + 6, pic - 6, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                std::vector<unsigned char> back(4096);
                CK(hipMemcpy(back.data(), dev, 4096, hipMemcpyDeviceToHost));
                printf("  first picture copied, bytes %s\n", memcmp(back.data(), m + 6, 4096) == 0 ? "equal" : "DIFFERENT");
                CK(hipHostUnregister(m));
            }
            t0 = now();
            hipError_t e = hipHostRegister(m, S, flags);
            t1 = now();
            printf("  hipHostRegister(flags %u) whole: %s, %.1f ms = %.1f GB/s\n", flags, hipGetErrorString(e), (t1 - t0) * 1e3, S / (t1 - t0) / 1e9);
            if (e != hipSuccess) { (void)hipGetLastError(); continue; }
            for (int rep = 0; rep < 2; ++rep) {
                t0 = now();
                for (size_t p = 0; p < npic; ++p) CK(hipMemcpyAsync(dev + (p % 64) * pic, m + p * pic + 6, pic - 6, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st)); t1 = now();
                printf("    copies of %zu pictures from the registered mapping: %.2f GB/s\n", npic, npic * pic / (t1 - t0) / 1e9);
            }
            // 2-D copies like the engine's (luma: pitch = width)
            t0 = now();
            for (size_t p = 0; p < npic; ++p) {
                CK(hipMemcpy2DAsync(dev + (p % 64) * dpic, 2048, m + p * pic + 6, 1920, 1920, 1080, hipMemcpyHostToDevice, st));
                CK(hipMemcpy2DAsync(dev + (p % 64) * dpic + 2048 * 1080, 1024, m + p * pic + 6 + 1920 * 1080, 960, 960, 1080, hipMemcpyHostToDevice, st));
            }
            CK(hipStreamSynchronize(st)); t1 = now();
            printf("    2-D copies (luma + chroma) from the registered mapping: %.2f GB/s\n", npic * pic / (t1 - t0) / 1e9);
            t0 = now(); CK(hipHostUnregister(m)); t1 = now();
            printf("    hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
        }
        // windows of 64 MB, registered one after the other
        {
            const size_t win = 64 << 20;
            t0 = now();
            size_t ok = 0;
            for (size_t o = 0; o + win <= S; o += win) if (hipHostRegister(m + o, win, hipHostRegisterReadOnly) == hipSuccess) ++ok; else (void)hipGetLastError();
            t1 = now();
            printf("  %zu windows of 64 MB registered: %.1f ms = %.1f GB/s\n", ok, (t1 - t0) * 1e3, ok * win / (t1 - t0) / 1e9);
            t0 = now();
            for (size_t o = 0; o + win <= S; o += win) (void)hipHostUnregister(m + o);
            t1 = now();
            printf("  unregistered: %.1f ms\n", (t1 - t0) * 1e3);
        }
        // two / four threads registering disjoint parts in windows of 64 MB at the same time
        for (int nt : {2, 4}) {
            const size_t win = 64 << 20, nwin = S / win;
            std::vector<std::thread> th;
            t0 = now();
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t i = t; i < nwin; i += nt) if (hipHostRegister(m + i * win, win, hipHostRegisterReadOnly) != hipSuccess) printf("register failed\n"); });
            for (auto &x : th) x.join();
            t1 = now();
            printf("  %d threads, windows of 64 MB: %.1f ms = %.1f GB/s\n", nt, (t1 - t0) * 1e3, nwin * win / (t1 - t0) / 1e9);
            for (size_t i = 0; i < nwin; ++i) (void)hipHostUnregister(m + i * win);
        }
        munmap(m, S);
    }
    // (b) today's path, one thread: pread into a page-locked ring, copies from there
    {
        unsigned char *ring; CK(hipHostMalloc(&ring, 64 * pic, 0));
        double t0 = now();
        for (size_t p = 0; p < npic; ++p) {
            if (p >= 64 && p % 64 == 0) CK(hipStreamSynchronize(st));
            if (pread(fd, ring + (p % 64) * pic, pic, p * pic) < 0) return 3;
            CK(hipMemcpyAsync(dev + (p % 64) * pic, ring + (p % 64) * pic, pic, hipMemcpyHostToDevice, st));
        }
        CK(hipStreamSynchronize(st)); double t1 = now();
        printf("pread into a page-locked ring (one thread) + copies: %.2f GB/s\n", npic * pic / (t1 - t0) / 1e9);
        t0 = now();
        for (int r = 0; r < 4; ++r) for (size_t p = 0; p < 64; ++p) CK(hipMemcpyAsync(dev + p * pic, ring + p * pic, pic, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st)); t1 = now();
        printf("copies from the page-locked ring alone: %.2f GB/s\n", 256 * pic / (t1 - t0) / 1e9);
        CK(hipHostFree(ring));
    }
    close(fd); unlink(path.c_str());
    return 0;
}
