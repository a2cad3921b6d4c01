// probe: what do returning 64-bit atomic adds cost on MI355X, as a function of the number of distinct counters and of the
// memory scope?  (the bucket scatter of the set build makes one reservation per run of k-mers, ~0.11 per input position,
// on ~6e5 bucket cursors; the level tables of phase 2 take one CAS per vertex)
//   agent scope     : what a plain atomicAdd compiles to -- executed at the memory side, coherent across the 8 XCDs
//   workgroup scope : executed in the L2 of the issuing XCD; only valid for counters that no other XCD touches
// build: hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
template <int SCOPE, bool XCD_PRIVATE>
__global__ void __launch_bounds__(256) adds(u64 *ctr, u64 n, int stride, int per_thread, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 x = mix(t + 7), s = 0;
    // XCD_PRIVATE: the counters are split into 8 regions, a workgroup only touches the region of the XCD it runs on
    u64 lo = 0, cnt = n;
    if (XCD_PRIVATE) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits 0-3
        cnt = n / 8; lo = (u64)(xcc & 7) * cnt;
    }
    for (int i = 0; i < per_thread; ++i) {
        x = mix(x + i);
        s += __hip_atomic_fetch_add(&ctr[(lo + x % cnt) * stride], 1ULL, __ATOMIC_RELAXED, SCOPE);
    }
    if (s == 42) out[0] = s;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    u64 *out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u64 nt = 4u << 20;
    const int per = 16;
    for (u64 n : {(u64)8, (u64)1024, (u64)65536, (u64)(600 * 1024), (u64)(16u << 20)}) {
        for (int stride : {1, 16}) {
            u64 *ctr; CK(hipMalloc(&ctr, n * stride * 8)); CK(hipMemset(ctr, 0, n * stride * 8));
            float ms[3];
            for (int v = 0; v < 3; ++v) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (v == 0) adds<__HIP_MEMORY_SCOPE_AGENT, false><<<(unsigned)(nt / 256), 256>>>(ctr, n, stride, per, out);
                    if (v == 1) adds<__HIP_MEMORY_SCOPE_WORKGROUP, false><<<(unsigned)(nt / 256), 256>>>(ctr, n, stride, per, out);
                    if (v == 2) adds<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<(unsigned)(nt / 256), 256>>>(ctr, n, stride, per, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                }
                CK(hipEventElapsedTime(&ms[v], e0, e1));
            }
            // check of the XCD-private variant: the counters must add up (L2 write-back at kernel end)
            CK(hipDeviceSynchronize());
            const double tot = (double)nt * per;
            printf("%9llu counters, stride %2d words: agent %6.2f G/s   workgroup scope %6.2f G/s   workgroup scope, XCD-private counters %6.2f G/s\n",
                   (unsigned long long)n, stride, tot / ms[0] / 1e6, tot / ms[1] / 1e6, tot / ms[2] / 1e6);
            CK(hipFree(ctr));
        }
    }
    // correctness of XCD-private workgroup-scope counting: sum of the counters == number of adds
    {
        const u64 n = 600 * 1024;
        u64 *ctr; CK(hipMalloc(&ctr, n * 8)); CK(hipMemset(ctr, 0, n * 8));
        adds<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<(unsigned)(nt / 256), 256>>>(ctr, n, 1, per, out);
        CK(hipDeviceSynchronize());
        static u64 h[600 * 1024];
        CK(hipMemcpy(h, ctr, n * 8, hipMemcpyDeviceToHost));
        u64 sum = 0; for (u64 i = 0; i < n; ++i) sum += h[i];
        printf("XCD-private workgroup-scope adds: counters sum to %llu, expected %llu\n", (unsigned long long)sum, (unsigned long long)(nt * per));
    }
    return 0;
}
