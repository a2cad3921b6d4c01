// Calibration of rocprofv3's FETCH_SIZE on gfx950 per ACCESS CLASS (round 6): the guide's "x2" is measured for wide coalesced
// streaming reads; the probe kernel's traffic is dominated by scattered lookups.  Every kernel here has a KNOWN number of reads of a
// known shape over a footprint far beyond the 256 MB Infinity Cache, so FETCH_SIZE / (known reads) = bytes tallied per read.
//   calib_stream16   : 16 B per lane, coalesced, every byte once
//   calib_stream8    : 8 B per lane, coalesced, every byte once
//   calib_scatter8   : independent aligned 8-byte reads at hashed addresses (one 64-byte sector each, almost surely distinct)
//   calib_tagwindow  : the probe kernel's pooled lookup: an UNALIGNED 8-byte read in a byte array (tags) + an aligned 8-byte read of
//                      the matching slot of a word array (keys) 8 times as large
//   calib_keywindow  : 64 contiguous bytes at an 8-byte aligned, hashed address (the "compare the 8 keys of the window" alternative)
//   calib_pair16     : one aligned 16-byte read at a hashed address (a node's two link words: the chain walks)
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin ; run under rocprofv3 --pmc (tools/fetch_calib.sh)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint64_t u64;
typedef uint8_t u8;
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__global__ void fill(u64 *a, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) a[i] = mix(i * 0x9e3779b97f4a7c15ULL + 1);
}
__global__ void __launch_bounds__(256) calib_stream16(const ulonglong2 *__restrict__ a, u64 n16, u64 *out) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) { ulonglong2 v = a[i]; s += v.x ^ v.y; }
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_stream8(const u64 *__restrict__ a, u64 n, u64 *out) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += a[i];
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_scatter8(const u64 *__restrict__ a, u64 n, int per_thread, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0, x = mix(t + 12345);
    for (int i = 0; i < per_thread; i += 4) {
        const u64 x0 = mix(x + 1), x1 = mix(x + 2), x2 = mix(x + 3), x3 = mix(x + 4);
        s += a[x0 % n] + a[x1 % n] + a[x2 % n] + a[x3 % n];
        x = x3;
    }
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_tagwindow(const u8 *__restrict__ tags, const u64 *__restrict__ keys, u64 nslots, int per_thread, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0, x = mix(t + 777);
    for (int i = 0; i < per_thread; i += 2) {
        const u64 x0 = mix(x + 1), x1 = mix(x + 2);
        const u64 s0 = x0 % (nslots - 8), s1 = x1 % (nslots - 8);
        u64 w0, w1;
        __builtin_memcpy(&w0, tags + s0, 8);
        __builtin_memcpy(&w1, tags + s1, 8);
        // the key of the slot the window points at (the tag value picks one of the 8: a dependent load, like the real lookup)
        s += keys[s0 + (w0 & 7)] + keys[s1 + (w1 & 7)];
        x = x1;
    }
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_keywindow(const u64 *__restrict__ keys, u64 nslots, int per_thread, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0, x = mix(t + 4242);
    for (int i = 0; i < per_thread; ++i) {
        x = mix(x + 1);
        const u64 *p = keys + x % (nslots - 8);
        const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(p);   // (8-byte aligned address, four 16-byte pieces: the compiler may split them)
        u64 v[8];
        __builtin_memcpy(v, p, 64);
        (void)q;
        s += v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
    }
    if (s == 42) out[0] = s;
}
__global__ void __launch_bounds__(256) calib_pair16(const ulonglong2 *__restrict__ a, u64 n16, int per_thread, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0, x = mix(t + 99);
    for (int i = 0; i < per_thread; i += 2) {
        const u64 x0 = mix(x + 1), x1 = mix(x + 2);
        const ulonglong2 v0 = a[x0 % n16], v1 = a[x1 % n16];
        s += v0.x + v0.y + v1.x + v1.y;
        x = x1;
    }
    if (s == 42) out[0] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const double gb = argc > 1 ? atof(argv[1]) : 40.0;       // footprint of the word array (the cfg[2] key table is 41 GB)
    const u64 n = (u64)(gb * 1e9) / 8;
    u64 *a, *out; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&out, 8));
    u8 *tags; CK(hipMalloc(&tags, n + 64));                   // one tag byte per word of `a`
    fill<<<65536, 256>>>(a, n);
    fill<<<65536, 256>>>(reinterpret_cast<u64 *>(tags), n / 8);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 nt = (u64)8192 * 1024;                         // threads of the scattered kernels
    const int per = 64;                                       // reads per thread -> 5.4e8 reads per launch
    float ms;
#define RUN(name, known_reads, bytes_each, ...) do { \
        CK(hipEventRecord(e0)); __VA_ARGS__; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("%-16s reads %.6e  useful bytes each %4d  -> %.3f GB useful, %.2f ms, %.1f G reads/s, %.0f GB/s useful\n", name, (double)(known_reads), \
               (int)(bytes_each), (double)(known_reads) * (bytes_each) / 1e9, ms, (double)(known_reads) / ms / 1e6, (double)(known_reads) * (bytes_each) / ms / 1e6); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN("calib_stream16", n / 2, 16, calib_stream16<<<256 * 32, 256>>>(reinterpret_cast<const ulonglong2 *>(a), n / 2, out));
        RUN("calib_stream8", n, 8, calib_stream8<<<256 * 32, 256>>>(a, n, out));
        RUN("calib_scatter8", nt * per, 8, calib_scatter8<<<(unsigned)(nt / 256), 256>>>(a, n, per, out));
        RUN("calib_tagwindow", nt * per, 16, calib_tagwindow<<<(unsigned)(nt / 256), 256>>>(tags, a, n, per, out));
        RUN("calib_keywindow", nt * per / 2, 64, calib_keywindow<<<(unsigned)(nt / 256), 256>>>(a, n, per / 2, out));
        RUN("calib_pair16", nt * per, 16, calib_pair16<<<(unsigned)(nt / 256), 256>>>(reinterpret_cast<const ulonglong2 *>(a), n / 2, per, out));
    }
    return 0;
}
