// How fast can a 2.55 GB output file on tmpfs get its pages, and be filled?  (round 5, host only; run on the GPU box.)  Not product code.
//   g++ -O2 -o tools/io_probe2.bin tools/io_probe2.cpp -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(int T, size_t n, const std::function<void(size_t, size_t)> &f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { f(n * t / T / 4096 * 4096, n * (t + 1) / T / 4096 * 4096); });
    f(0, n / T / 4096 * 4096);
    for (auto &x : th) x.join();
}
static const char *PATH = "/dev/shm/kc_io_probe2.out";
static int fresh(size_t N, bool size_it) {
    unlink(PATH);
    int fd = open(PATH, O_RDWR | O_CREAT, 0644);
    if (size_it && ftruncate(fd, (off_t)N) != 0) printf("ftruncate failed\n");
    return fd;
}
static void cat(const char *p) {
    FILE *f = fopen(p, "r");
    if (!f) { printf("%s: (absent)\n", p); return; }
    char buf[512];
    while (fgets(buf, sizeof buf, f)) printf("%s: %s", p, buf);
    fclose(f);
}

int main(int argc, char **argv) {
    const size_t N = (size_t)((argc > 1 ? atof(argv[1]) : 2.55) * 1e9) / 4096 * 4096;
    cat("/sys/kernel/mm/transparent_hugepage/shmem_enabled");
    cat("/sys/kernel/mm/transparent_hugepage/enabled");
    { FILE *f = fopen("/proc/mounts", "r"); char b[1024]; while (f && fgets(b, sizeof b, f)) if (strstr(b, "/dev/shm") || strstr(b, " /tmp ")) printf("mount: %s", b); if (f) fclose(f); }
    { FILE *f = popen("uname -r; nproc; numactl -H 2>/dev/null | head -4", "r"); char b[512]; while (f && fgets(b, sizeof b, f)) printf("sys: %s", b); if (f) pclose(f); }
    char *src = (char *)malloc(N);
    par(8, N, [&](size_t lo, size_t hi) { memset(src + lo, 'C', hi - lo); });

    // E. populate with T threads, no fallocate
    for (int T : {1, 2, 4, 8}) {
        int fd = fresh(N, true);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double a = now();
        par(T, N, [&](size_t lo, size_t hi) { if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0; });
        double b = now();
        par(8, N, [&](size_t lo, size_t hi) { memcpy(m + lo, src + lo, hi - lo); });
        double c = now();
        munmap(m, N);
        double d = now();
        close(fd);
        printf("[populate x%d] %.1f ms (%.1f GB/s); memcpy x8 into it %.1f ms; munmap %.1f ms\n", T, (b - a) * 1e3, N / (b - a) / 1e9, (c - b) * 1e3, (d - c) * 1e3);
    }
    // A/B. fallocate, then populate x8, memcpy, munmap
    {
        int fd = fresh(N, false);
        double a = now();
        int rc = fallocate(fd, 0, 0, (off_t)N);
        double b = now();
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par(8, N, [&](size_t lo, size_t hi) { if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0; });
        double c = now();
        par(8, N, [&](size_t lo, size_t hi) { memcpy(m + lo, src + lo, hi - lo); });
        double d = now();
        munmap(m, N);
        double e = now();
        close(fd);
        printf("[fallocate] rc=%d %.1f ms (%.1f GB/s); populate x8 after it %.1f ms; memcpy x8 %.1f ms; munmap %.1f ms\n", rc, (b - a) * 1e3, N / (b - a) / 1e9, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3);
    }
    // fallocate in T parallel ranges
    for (int T : {2, 4}) {
        int fd = fresh(N, false);
        double a = now();
        par(T, N, [&](size_t lo, size_t hi) { if (fallocate(fd, 0, (off_t)lo, (off_t)(hi - lo)) != 0) printf("fallocate failed\n"); });
        double b = now();
        close(fd);
        printf("[fallocate x%d ranges] %.1f ms\n", T, (b - a) * 1e3);
    }
    // fallocate, then memcpy x8 straight into the unpopulated mapping (faults zero + map), munmap
    for (int T : {8, 16}) {
        int fd = fresh(N, false);
        if (fallocate(fd, 0, 0, (off_t)N) != 0) printf("fallocate failed\n");
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double c = now();
        par(T, N, [&](size_t lo, size_t hi) { memcpy(m + lo, src + lo, hi - lo); });
        double d = now();
        munmap(m, N);
        double e = now();
        close(fd);
        printf("[fallocate, then memcpy x%d into the unpopulated mapping] %.1f ms (%.1f GB/s); munmap %.1f ms\n", T, (d - c) * 1e3, N / (d - c) / 1e9, (e - d) * 1e3);
    }
    // C. fallocate, then pwrite with T threads (32 MB pieces interleaved)
    for (int T : {1, 2, 4, 8}) {
        int fd = fresh(N, false);
        if (fallocate(fd, 0, 0, (off_t)N) != 0) printf("fallocate failed\n");
        double a = now();
        par(T, N, [&](size_t lo, size_t hi) { size_t p = lo; while (p < hi) { size_t n = hi - p < ((size_t)8 << 20) ? hi - p : ((size_t)8 << 20); ssize_t w = pwrite(fd, src + p, n, (off_t)p); if (w <= 0) break; p += (size_t)w; } });
        double b = now();
        close(fd);
        printf("[fallocate, then pwrite x%d] %.1f ms (%.1f GB/s)\n", T, (b - a) * 1e3, N / (b - a) / 1e9);
    }
    // D. allocating pwrite with T threads, no fallocate
    for (int T : {1, 2, 4}) {
        int fd = fresh(N, false);
        double a = now();
        par(T, N, [&](size_t lo, size_t hi) { size_t p = lo; while (p < hi) { size_t n = hi - p < ((size_t)8 << 20) ? hi - p : ((size_t)8 << 20); ssize_t w = pwrite(fd, src + p, n, (off_t)p); if (w <= 0) break; p += (size_t)w; } });
        double b = now();
        close(fd);
        printf("[allocating pwrite x%d] %.1f ms (%.1f GB/s)\n", T, (b - a) * 1e3, N / (b - a) / 1e9);
    }
    // F. huge pages on the mapping, if the mount / sysfs setting allows
    {
        int fd = fresh(N, true);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        int rc = madvise(m, N, MADV_HUGEPAGE);
        double a = now();
        par(8, N, [&](size_t lo, size_t hi) { if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0; });
        double b = now();
        munmap(m, N);
        double c = now();
        close(fd);
        printf("[MADV_HUGEPAGE rc=%d] populate x8 %.1f ms, munmap %.1f ms\n", rc, (b - a) * 1e3, (c - b) * 1e3);
    }
    // G. several files (one per thread): is the contention per inode?
    {
        const int T = 8;
        std::vector<int> fds(T);
        std::vector<char *> ms(T);
        const size_t per = N / T / 4096 * 4096;
        for (int t = 0; t < T; ++t) { char p[64]; snprintf(p, sizeof p, "/dev/shm/kc_io_probe2.%d", t); unlink(p); fds[t] = open(p, O_RDWR | O_CREAT, 0644); if (ftruncate(fds[t], (off_t)per) != 0) {} ms[t] = (char *)mmap(nullptr, per, PROT_READ | PROT_WRITE, MAP_SHARED, fds[t], 0); }
        double a = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { if (madvise(ms[t], per, MADV_POPULATE_WRITE) != 0) for (size_t p = 0; p < per; p += 4096) ms[t][p] = 0; });
        for (auto &x : th) x.join();
        double b = now();
        for (int t = 0; t < T; ++t) { munmap(ms[t], per); close(fds[t]); char p[64]; snprintf(p, sizeof p, "/dev/shm/kc_io_probe2.%d", t); unlink(p); }
        printf("[8 files, one populating thread each] %.1f ms (%.1f GB/s)\n", (b - a) * 1e3, per * T / (b - a) / 1e9);
    }
    // H. anonymous memory for comparison (is it tmpfs or page allocation in general?)
    {
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(m, N, MADV_NOHUGEPAGE);
        double a = now();
        par(8, N, [&](size_t lo, size_t hi) { if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0; });
        double b = now();
        munmap(m, N);
        printf("[anonymous 4K pages] populate x8 %.1f ms, munmap %.1f ms\n", (b - a) * 1e3, (now() - b) * 1e3);
    }
    unlink(PATH);
    return 0;
}
