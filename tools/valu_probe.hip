// Issue cost of the integer VALU instructions the probe / build kernels are made of, on gfx950: every wave runs ITER rounds of
// 8 independent chains of one instruction; with all SIMDs full the time gives cycles per wave-instruction (4 = full rate).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe.bin ; run: tools/valu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32;
typedef uint64_t u64;
#define ITER 4096

#define KERNEL32(name, ASM)                                                                         \
    __global__ void __launch_bounds__(256) name(u32 *out, u32 seed) {                               \
        u32 a[8];                                                                                   \
        for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 9 + i;                              \
        u32 b = seed | 1u;                                                                          \
        for (int it = 0; it < ITER; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b)); \
        }                                                                                           \
        u32 s = 0;                                                                                  \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
    }
#define KERNEL64(name, ASM)                                                                         \
    __global__ void __launch_bounds__(256) name(u64 *out, u32 seed) {                               \
        u64 a[8];                                                                                   \
        for (int i = 0; i < 8; ++i) a[i] = ((u64)seed << 20) + threadIdx.x * 9 + i;                 \
        u32 b = (seed & 7u) | 1u;                                                                   \
        for (int it = 0; it < ITER; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b)); \
        }                                                                                           \
        u64 s = 0;                                                                                  \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                             \
    }

KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")
KERNEL32(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL32(k_bfrev, "v_bfrev_b32 %0, %0")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %0")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %0")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %0")
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_add64, "v_lshl_add_u64 %0, %0, 0, %0")

__global__ void __launch_bounds__(256) k_cmp64(u64 *out, u32 seed) {   // 64-bit compare + select, as the compiler writes it
    u64 a[8];
    for (int i = 0; i < 8; ++i) a[i] = ((u64)seed << 20) + threadIdx.x * 9 + i;
    u64 b = ((u64)seed << 19) + 77;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = a[i] < b ? a[i] + 3 : b + (u64)it; asm volatile("" : "+v"(a[i])); }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename T, typename K> static void run(const char *name, K kern, int per_iter) {
    T *out;
    hipMalloc(&out, (size_t)256 * 2048 * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * ITER * 8 * per_iter / (256 CUs * 4 SIMDs)
    const double wi = (double)blocks * 4 * ITER * 8 * per_iter / (256.0 * 4);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("%-12s %8.3f ms  %6.2f cycles per wave-instruction (at %d MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / wi, clk / 1000);
    hipFree(out);
}

int main() {
    run<u32>("xor", k_xor, 1); run<u32>("add", k_add, 1); run<u32>("mul_lo", k_mul_lo, 1); run<u32>("mul_hi", k_mul_hi, 1);
    run<u32>("mul_u24", k_mul24, 1); run<u32>("mad_u24", k_mad24, 1); run<u32>("alignbit", k_alignbit, 1); run<u32>("bfe", k_bfe, 1);
    run<u32>("min_u32", k_min, 1); run<u32>("ffbl", k_ffbl, 1); run<u32>("bfrev", k_bfrev, 1); run<u32>("bcnt", k_bcnt, 1);
    run<u32>("lshl_add", k_lshl_add, 1); run<u32>("xad", k_xad, 1); run<u32>("and_or", k_and_or, 1); run<u32>("perm", k_perm, 1);
    run<u64>("lshl_b64", k_lshl64, 1); run<u64>("lshr_b64", k_lshr64, 1); run<u64>("mad_u64_u32", k_mad64, 1); run<u64>("lshl_add_u64", k_add64, 1);
    run<u64>("cmp64+sel", k_cmp64, 1);
    return 0;
}
