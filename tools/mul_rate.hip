// Issue rate of the integer multiplies the hashes are made of, against an add (round 5; run on the GPU box): hipcc -O3 --offload-arch=gfx950 -o tools/mul_rate.bin tools/mul_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32; typedef unsigned long long u64;
template <int OP> __global__ void __launch_bounds__(256) k(u32 *out, int n, u32 seed) {
    u32 a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
    u64 b[4];
    for (int i = 0; i < 4; ++i) b[i] = ((u64)seed << 32) | (threadIdx.x * 4 + i);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + (a[(i + 1) & 7] ^ 0x9E3779B1u);          // add + xor: 2 full-rate instructions
            if (OP == 1) a[i] = a[i] * 0x9E3779B1u + 1u;                         // v_mul_lo_u32 (+ add)
            if (OP == 2) a[i] = __umulhi(a[i], 0x9E3779B1u) + a[(i + 1) & 7];    // v_mul_hi_u32 (+ add)
            if (OP == 3) a[i] = __umul24(a[i], 0x9E3779u) + 1u;                  // v_mul_u32_u24 (+ add)
        }
        if (OP == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = b[i] * 0xff51afd7ed558ccdULL + 1;  // 64-bit multiply
        }
    }
    u32 r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    for (int i = 0; i < 4; ++i) r ^= (u32)b[i] ^ (u32)(b[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> double run(u32 *d, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 32, 256>>>(d, 16, 1);
    hipEventRecord(e0);
    k<OP><<<256 * 32, 256>>>(d, n, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    u32 *d; hipMalloc(&d, 256 * 32 * 256 * 4);
    const int n = 4096;
    const double lanes = 256.0 * 32 * 256;
    const char *names[] = {"add+xor (2 instr)", "mul_lo_u32 + add", "mul_hi_u32 + add", "mul_u32_u24 + add", "64-bit mul + add (x4 per trip)"};
    double t[5] = {run<0>(d, n), run<1>(d, n), run<2>(d, n), run<3>(d, n), run<4>(d, n)};
    for (int i = 0; i < 5; ++i) printf("%-32s %.3f ms  = %.2f T lane-ops/s (statements)\n", names[i], t[i], lanes * n * (i == 4 ? 4 : 8) / t[i] / 1e9);
    return 0;
}
