// probe: what does MI355X sustain for scattered 8-byte reads, as a function of the footprint and of the access pattern?
// (the two dominant kernels of the path -- neighbour lookup and chain walk -- are scattered-read kernels; this gives their
// ceilings independently of the k-mer code)
//   indep      : every thread issues independent loads at hashed addresses over the whole footprint (4 in flight per thread)
//   chase      : every thread follows a dependent chain of random addresses over the whole footprint (the hop walk's pattern)
//   chase_near : a dependent chain that takes 9 steps inside a 128 KB region, then jumps to a random region (the plain
//                chain walk's pattern: ~90 % of the steps stay in the table block)
// build: hipcc --offload-arch=gfx950 -O3 tools/rand_probe.hip -o tools/rand_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__global__ void fill(u64 *a, u64 n) {
    u64 i = (u64)blockIdx.y * gridDim.x * blockDim.x + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = mix(i * 0x9e3779b97f4a7c15ULL + 1);
}
__global__ void __launch_bounds__(256) indep(const u64 *__restrict__ a, u64 n, int steps, u64 *out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0, x = mix(t + 12345);
    for (int i = 0; i < steps; i += 4) {
        u64 x0 = mix(x + 1), x1 = mix(x + 2), x2 = mix(x + 3), x3 = mix(x + 4);
        s += a[x0 % n] + a[x1 % n] + a[x2 % n] + a[x3 % n];
        x = x3;
    }
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) chase(const u64 *__restrict__ a, u64 n, int steps, u64 *out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 p = mix(t + 999) % n;
    for (int i = 0; i < steps; ++i) p = a[p] % n;
    if (p == ~0ull) out[0] = p;
}
__global__ void __launch_bounds__(256) chase_near(const u64 *__restrict__ a, u64 n, int steps, int near, u64 *out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 p = mix(t + 999) % n;
    const u64 region = 16384;                      // 128 KB of 8-byte words
    for (int i = 0; i < steps; ++i) {
        const u64 v = a[p];
        if (i % (near + 1) == near) p = v % n;
        else p = (p & ~(region - 1)) | (v & (region - 1));
    }
    if (p == ~0ull) out[0] = p;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    u64 *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gbs[] = {0.125, 1, 4, 16, 40, 80};
    for (double gb : gbs) {
        const u64 n = (u64)(gb * (1ull << 30)) / 8;
        u64 *a; CK(hipMalloc(&a, n * 8));
        dim3 g((unsigned)65535, (unsigned)((n + 65535ull * 256 - 1) / (65535ull * 256)));
        fill<<<g, 256>>>(a, n); CK(hipDeviceSynchronize());
        for (int threads_k : {512, 2048, 8192}) {
            const u64 nt = (u64)threads_k * 1024;
            const int steps = 256;
            float ms;
            for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); indep<<<(unsigned)(nt / 256), 256>>>(a, n, steps, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double gi = (double)nt * steps / ms / 1e6;
            for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); chase<<<(unsigned)(nt / 256), 256>>>(a, n, steps, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double gc = (double)nt * steps / ms / 1e6;
            for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(e0)); chase_near<<<(unsigned)(nt / 256), 256>>>(a, n, steps, 9, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double gn = (double)nt * steps / ms / 1e6;
            printf("footprint %6.3f GB  threads %5dk : indep %6.1f G loads/s   chase %6.1f   chase_near(9 in 128 KB + 1 far) %6.1f\n", gb, threads_k, gi, gc, gn);
        }
        CK(hipFree(a));
    }
    return 0;
}
