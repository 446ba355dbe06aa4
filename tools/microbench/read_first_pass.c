// read_first_pass: how fast can N threads bring a FRESHLY WRITTEN tmpfs file into a buffer -- pread of 2-MB pieces (what the CLI's
// readers do), memcpy out of a shared mapping, memcpy after MADV_POPULATE_READ of the piece -- on the first pass over the file and on
// the second?  (The CLI's first pass over a new clip runs at a third of the later ones.)
// build: gcc -O2 -pthread -o read_first_pass read_first_pass.c     usage: read_first_pass <file> <threads> <pread|mmap|populate> [passes]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static int fd, mode;
static size_t size, piece = 2u << 20;
static unsigned char *map;
static atomic_size_t next_off;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *work(void *arg)
{
    unsigned char *buf = aligned_alloc(4096, piece);
    memset(buf, 1, piece);
    (void)arg;
    for (;;) {
        size_t off = atomic_fetch_add(&next_off, piece);
        if (off >= size) break;
        size_t n = size - off < piece ? size - off : piece;
        if (mode == 0) { size_t d = 0; while (d < n) { ssize_t g = pread(fd, buf + d, n - d, (off_t)(off + d)); if (g <= 0) exit(3); d += (size_t)g; } }
        else {
            if (mode == 2) madvise(map + off, n, MADV_POPULATE_READ);
            memcpy(buf, map + off, n);
        }
    }
    free(buf);
    return NULL;
}
int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int nt = atoi(argv[2]), passes = argc > 4 ? atoi(argv[4]) : 2;
    mode = !strcmp(argv[3], "pread") ? 0 : !strcmp(argv[3], "mmap") ? 1 : 2;
    fd = open(argv[1], O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st)) return 1;
    size = (size_t)st.st_size;
    for (int p = 0; p < passes; ++p) {
        double t0 = now();
        if (mode) { map = mmap(NULL, size, PROT_READ, MAP_SHARED, fd, 0); if (map == MAP_FAILED) return 1; }
        atomic_store(&next_off, 0);
        pthread_t th[64];
        for (int t = 0; t < nt; ++t) pthread_create(&th[t], NULL, work, NULL);
        for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
        if (mode) munmap(map, size);
        double dt = now() - t0;
        printf("%-8s %2d threads, pass %d: %5.1f GB/s\n", argv[3], nt, p + 1, size / dt / 1e9);
    }
    return 0;
}
