// mmap_window_probe: the cost of page-locking a clip window by window -- every window its own mmap() of a range of the file (fresh
// addresses: nothing the runtime or the kernel could have kept from an earlier registration), registered read-only, with 1 / 2 / 4 / 8
// threads working on different windows at the same time; then linear copies of 3-MB pictures out of the windows, then unregister +
// munmap.  usage: mmap_window_probe <dir> <file MB> <window MB>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <atomic>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t S = (size_t)(argc > 2 ? atoll(argv[2]) : 4096) << 20, win = (size_t)(argc > 3 ? atoll(argv[3]) : 192) << 20, nwin = S / win;
    const size_t pic = 3110400;
    const std::string path = dir + "/mmap_window_probe.bin";
    {
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<unsigned char> buf(8 << 20);
        for (size_t i = 0; i < buf.size(); ++i) buf[i] = (unsigned char)(i * 2654435761u >> 13);
        for (size_t done = 0; done < S; done += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) return 2;
        close(fd);
    }
    if (hipSetDevice(0) != hipSuccess) return 1;
    unsigned char *dev;
    if (hipMalloc(&dev, 64 * pic) != hipSuccess) return 1;
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int fd = open(path.c_str(), O_RDONLY);
    for (int pass = 0; pass < 2; ++pass)
        for (int nt : {1, 2, 4, 8}) {
            std::vector<unsigned char *> m(nwin, nullptr);
            std::atomic<size_t> next{0};
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            double t0 = now();
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    (void)hipSetDevice(0);
                    for (size_t i; (i = next.fetch_add(1)) < nwin;) {
                        unsigned char *p = (unsigned char *)mmap(nullptr, win, PROT_READ, MAP_SHARED, fd, (off_t)(i * win));
                        if (p == MAP_FAILED || hipHostRegister(p, win, hipHostRegisterReadOnly) != hipSuccess) { ++bad; (void)hipGetLastError(); continue; }
                        m[i] = p;
                    }
                });
            for (auto &x : th) x.join();
            double t1 = now();
            printf("pass %d, %d thread(s): %zu windows of %zu MB mapped + registered in %.1f ms = %.1f GB/s (%d failed)\n", pass, nt, nwin, win >> 20, (t1 - t0) * 1e3,
                   nwin * win / (t1 - t0) / 1e9, bad.load());
            if (nt == 1 || nt == 8) {
                t0 = now();
                size_t n = 0;
                for (size_t i = 0; i < nwin; ++i)
                    for (size_t o = 0; m[i] && o + pic <= win; o += pic, ++n) (void)hipMemcpyAsync(dev + (n % 64) * pic, m[i] + o, pic, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);
                t1 = now();
                printf("    %zu linear copies of 3 MB out of the windows: %.2f GB/s\n", n, n * pic / (t1 - t0) / 1e9);
            }
            t0 = now();
            for (size_t i = 0; i < nwin; ++i) if (m[i]) { (void)hipHostUnregister(m[i]); munmap(m[i], win); }
            t1 = now();
            printf("    unregistered + unmapped in %.1f ms\n", (t1 - t0) * 1e3);
        }
    close(fd); unlink(path.c_str());
    return 0;
}
