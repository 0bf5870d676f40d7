// tools/mb_touch.c -- what does the first touch of a fresh 4 GB anonymous mapping cost on this host, with 16 threads:
// 4 KB pages, MADV_HUGEPAGE, MADV_POPULATE_WRITE.  gcc -O2 -pthread tools/mb_touch.c -o tools/mb_touch.bin
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static char *base; static size_t bytes; static int nthr, mode;
static void *work(void *arg) {
    size_t i = (size_t)(intptr_t)arg, per = (bytes / nthr) & ~(size_t)((2u << 20) - 1), lo = i * per, hi = (i + 1 == (size_t)nthr) ? bytes : lo + per;
    if (mode == 2) { if (madvise(base + lo, hi - lo, MADV_POPULATE_WRITE)) perror("populate"); }
    else if (mode == 3) memset(base + lo, 1, hi - lo);
    else for (size_t o = lo; o < hi; o += 4096) base[o] = 1;
    return 0;
}
int main(int argc, char **argv) {
    bytes = (size_t)4 << 30; nthr = argc > 1 ? atoi(argv[1]) : 16;
    const char *names[] = {"touch 4K", "MADV_HUGEPAGE + touch", "MADV_POPULATE_WRITE", "memset", "MADV_HUGEPAGE + memset"};
    for (int m = 0; m < 5; ++m) {
        base = mmap(0, bytes + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base == MAP_FAILED) return 1;
        char *al = (char *)(((uintptr_t)base + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
        char *keep = base; base = al;
        if (m == 1 || m == 4) madvise(base, bytes, MADV_HUGEPAGE);
        mode = (m == 4) ? 3 : (m == 1 ? 0 : m);
        pthread_t t[64]; double t0 = now();
        for (int i = 0; i < nthr; ++i) pthread_create(&t[i], 0, work, (void *)(intptr_t)i);
        for (int i = 0; i < nthr; ++i) pthread_join(t[i], 0);
        double dt = now() - t0;
        printf("%-26s %2d threads  %7.1f ms  %6.1f GB/s\n", names[m], nthr, dt * 1e3, bytes / dt / 1e9);
        munmap(keep, bytes + (2u << 20));
    }
    return 0;
}
