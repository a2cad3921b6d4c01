// Host <-> device I/O options for the CLI's end-to-end second, measured on the GPU box (round 5).  Not product code.
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/io_probe.bin tools/io_probe.hip -lpthread
//   run:   tools/io_probe.bin [GB]        (files in /dev/shm)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <functional>
#include <unistd.h>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at line %d: %s\n", #x, __LINE__, hipGetErrorString(e__)); } } while (0)

static void par(int T, size_t n, const std::function<void(size_t, size_t)> &f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { f(n * t / T, n * (t + 1) / T); });
    f(0, n / T);
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.6;
    const size_t N = (size_t)(gb * 1e9) / 4096 * 4096;
    double t0 = now();
    CK(hipInit(0));
    double t1 = now();
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double t2 = now();
    char *d = nullptr;
    CK(hipMalloc((void **)&d, N));
    CK(hipMemsetAsync(d, 'A', N, s));
    CK(hipStreamSynchronize(s));
    double t3 = now();
    printf("hipInit %.1f ms, device+stream %.1f ms, hipMalloc+memset %.2f GB %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, N / 1e9, (t3 - t2) * 1e3);

    // ---- 1. ceiling: D2H / H2D with a big pinned buffer
    {
        char *h = nullptr;
        double a = now();
        CK(hipHostMalloc((void **)&h, N, hipHostMallocDefault));
        double b = now();
        CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double c = now();
        CK(hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
        double e = now();
        CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double f = now();
        printf("[pinned] hipHostMalloc %.1f ms; D2H first %.1f ms (%.1f GB/s), H2D %.1f ms (%.1f GB/s), D2H again %.1f ms (%.1f GB/s)\n", (b - a) * 1e3, (c - b) * 1e3,
               N / (c - b) / 1e9, (e - c) * 1e3, N / (e - c) / 1e9, (f - e) * 1e3, N / (f - e) / 1e9);
        // CPU memcpy out of pinned memory into malloc'ed memory, T threads
        char *m = (char *)malloc(N);
        memset(m, 1, N);
        for (int T : {1, 4, 8, 16, 32}) {
            double g = now();
            par(T, N, [&](size_t lo, size_t hi) { memcpy(m + lo, h + lo, hi - lo); });
            double hh = now();
            par(T, N, [&](size_t lo, size_t hi) { memcpy(h + lo, m + lo, hi - lo); });
            double ii = now();
            printf("[pinned] memcpy pinned->malloc T=%d: %.1f ms (%.1f GB/s); malloc->pinned %.1f ms (%.1f GB/s)\n", T, (hh - g) * 1e3, N / (hh - g) / 1e9, (ii - hh) * 1e3,
                   N / (ii - hh) / 1e9);
        }
        free(m);
        double g = now();
        CK(hipHostFree(h));
        printf("[pinned] hipHostFree %.1f ms\n", (now() - g) * 1e3);
    }

    // ---- 2. output file on tmpfs: map, populate, register, D2H straight into the page cache
    for (int variant = 0; variant < 3; ++variant) {
        const char *path = "/dev/shm/kc_io_probe.out";
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT, 0644);
        double a = now();
        if (ftruncate(fd, (off_t)N) != 0) printf("ftruncate failed\n");
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double b = now();
        if (variant >= 1) {   // populate first with 8 / 32 threads
            const int T = variant == 1 ? 8 : 32;
            par(T, N, [&](size_t lo, size_t hi) {
                lo = lo / 4096 * 4096; hi = hi / 4096 * 4096;
                if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0;
            });
        }
        double c = now();
        hipError_t e = hipHostRegister(m, N, hipHostRegisterDefault);
        double dd = now();
        printf("[out v%d] ftruncate+mmap %.1f ms, populate %.1f ms, hipHostRegister %.1f ms (%s)\n", variant, (b - a) * 1e3, (c - b) * 1e3, (dd - c) * 1e3, hipGetErrorString(e));
        if (e == hipSuccess) {
            CK(hipMemcpyAsync(m, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            double ee = now();
            CK(hipHostUnregister(m));
            double ff = now();
            printf("[out v%d] D2H into the mapping %.1f ms (%.1f GB/s), unregister %.1f ms\n", variant, (ee - dd) * 1e3, N / (ee - dd) / 1e9, (ff - ee) * 1e3);
        } else (void)hipGetLastError();
        double g = now();
        munmap(m, N);
        double h = now();
        if (ftruncate(fd, (off_t)(N - 12345)) != 0) printf("ftruncate failed\n");
        close(fd);
        double i = now();
        printf("[out v%d] munmap %.1f ms, ftruncate+close %.1f ms\n", variant, (h - g) * 1e3, (i - h) * 1e3);
        unlink(path);
    }
    // chunked registration (what a background thread would do while the GPU computes): 64 MB pieces
    {
        const char *path = "/dev/shm/kc_io_probe.out";
        int fd = open(path, O_RDWR | O_CREAT, 0644);
        if (ftruncate(fd, (off_t)N) != 0) printf("ftruncate failed\n");
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        const size_t P = (size_t)256 << 20;
        double a = now();
        size_t ok = 0;
        for (size_t lo = 0; lo < N; lo += P) { if (hipHostRegister(m + lo, lo + P < N ? P : N - lo, hipHostRegisterDefault) == hipSuccess) ++ok; }
        double b = now();
        for (size_t lo = 0; lo < N; lo += P) CK(hipMemcpyAsync(m + lo, d + lo, lo + P < N ? P : N - lo, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double c = now();
        for (size_t lo = 0; lo < N; lo += P) (void)hipHostUnregister(m + lo);
        double e = now();
        printf("[out chunks 256MB] register (unpopulated) %.1f ms (%zu ok), D2H %.1f ms (%.1f GB/s), unregister %.1f ms\n", (b - a) * 1e3, ok, (c - b) * 1e3, N / (c - b) / 1e9, (e - c) * 1e3);
        munmap(m, N); close(fd);
        // the file stays for the input test below
    }
    // ---- 3. input file on tmpfs: map read-only, register, H2D straight from the page cache; against pread into pinned chunks
    {
        const char *path = "/dev/shm/kc_io_probe.out";
        int fd = open(path, O_RDONLY);
        double a = now();
        char *m = (char *)mmap(nullptr, N, PROT_READ, MAP_SHARED, fd, 0);
        hipError_t e = hipHostRegister(m, N, hipHostRegisterReadOnly);
        double b = now();
        if (e != hipSuccess) { (void)hipGetLastError(); printf("[in] hipHostRegisterReadOnly: %s; trying default flags\n", hipGetErrorString(e)); e = hipHostRegister(m, N, hipHostRegisterDefault); b = now(); }
        printf("[in] mmap + hipHostRegister %.1f ms (%s)\n", (b - a) * 1e3, hipGetErrorString(e));
        if (e == hipSuccess) {
            CK(hipMemcpyAsync(d, m, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
            double c = now();
            CK(hipHostUnregister(m));
            printf("[in] H2D from the mapping %.1f ms (%.1f GB/s), unregister %.1f ms\n", (c - b) * 1e3, N / (c - b) / 1e9, (now() - c) * 1e3);
        } else (void)hipGetLastError();
        munmap(m, N);
        // plain hipMemcpy from the unregistered mapping (the runtime's pageable path)
        m = (char *)mmap(nullptr, N, PROT_READ, MAP_SHARED, fd, 0);
        double g = now();
        CK(hipMemcpy(d, m, N, hipMemcpyHostToDevice));
        double h = now();
        printf("[in] hipMemcpy from the unregistered mapping %.1f ms (%.1f GB/s)\n", (h - g) * 1e3, N / (h - g) / 1e9);
        munmap(m, N);
        // pread into pinned chunks with T threads (32 MB chunks, 3 buffers, as kc_read_fasta_device), H2D per chunk
        for (int T : {4, 8, 16}) {
            const size_t CH = (size_t)32 << 20;
            char *hb[3];
            hipEvent_t ev[3];
            for (int i = 0; i < 3; ++i) { CK(hipHostMalloc((void **)&hb[i], CH, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
            double a2 = now();
            int b3 = 0;
            size_t nchunk = 0;
            for (size_t lo = 0; lo < N; lo += CH, b3 = (b3 + 1) % 3, ++nchunk) {
                const size_t n = lo + CH < N ? CH : N - lo;
                if (nchunk >= 3) CK(hipEventSynchronize(ev[b3]));
                par(T, n, [&](size_t x, size_t y) { size_t p = x; while (p < y) { ssize_t r = pread(fd, hb[b3] + p, y - p, (off_t)(lo + p)); if (r <= 0) break; p += (size_t)r; } });
                CK(hipMemcpyAsync(d + lo, hb[b3], n, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[b3], s));
            }
            CK(hipStreamSynchronize(s));
            double b2 = now();
            printf("[in] pread x%d into pinned 32 MB chunks + H2D: %.1f ms (%.1f GB/s)\n", T, (b2 - a2) * 1e3, N / (b2 - a2) / 1e9);
            for (int i = 0; i < 3; ++i) { CK(hipHostFree(hb[i])); CK(hipEventDestroy(ev[i])); }
        }
        close(fd);
        unlink(path);
    }
    // ---- 4. output through pinned chunks: memcpy into a populated mapping (today's path) with T threads
    {
        const char *path = "/dev/shm/kc_io_probe.out";
        int fd = open(path, O_RDWR | O_CREAT, 0644);
        if (ftruncate(fd, (off_t)N) != 0) printf("ftruncate failed\n");
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double a = now();
        par(8, N, [&](size_t lo, size_t hi) { lo = lo / 4096 * 4096; hi = hi / 4096 * 4096; if (madvise(m + lo, hi - lo, MADV_POPULATE_WRITE) != 0) for (size_t p = lo; p < hi; p += 4096) m[p] = 0; });
        printf("[out pinned] populate x8 %.1f ms\n", (now() - a) * 1e3);
        for (int T : {4, 16}) {
            const size_t CH = (size_t)32 << 20;
            char *hb[2];
            hipEvent_t ev[2];
            for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void **)&hb[i], CH, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
            double a2 = now();
            size_t sent = 0, done = 0;
            int slot = 0;
            size_t len[2] = {CH < N ? CH : N, 0};
            CK(hipMemcpyAsync(hb[0], d, len[0], hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[0], s));
            sent = len[0];
            while (done < N) {
                const int nxt = slot ^ 1;
                if (sent < N) { len[nxt] = N - sent < CH ? N - sent : CH; CK(hipMemcpyAsync(hb[nxt], d + sent, len[nxt], hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[nxt], s)); sent += len[nxt]; }
                CK(hipEventSynchronize(ev[slot]));
                par(T, len[slot], [&](size_t x, size_t y) { memcpy(m + done + x, hb[slot] + x, y - x); });
                done += len[slot];
                slot = nxt;
            }
            double b2 = now();
            printf("[out pinned] D2H chunks + memcpy x%d into the populated mapping: %.1f ms (%.1f GB/s)\n", T, (b2 - a2) * 1e3, N / (b2 - a2) / 1e9);
            for (int i = 0; i < 2; ++i) { CK(hipHostFree(hb[i])); CK(hipEventDestroy(ev[i])); }
        }
        munmap(m, N); close(fd); unlink(path);
    }
    double z = now();
    CK(hipFree(d));
    printf("hipFree %.1f ms\n", (now() - z) * 1e3);
    return 0;
}
