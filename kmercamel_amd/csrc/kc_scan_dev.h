// Wavefront / workgroup prefix sums (64-wide wavefronts, 256-thread workgroups) shared by the kernels that compact or lay out lists.
#pragma once
#include "kc_key.h"
#if defined(__HIPCC__)
__device__ __forceinline__ u64 kc_wave_incl_scan(u64 v) {   // inclusive scan over the 64 lanes of a wavefront
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
// exclusive scan of one value per thread over a 256-thread workgroup; returns the exclusive prefix, *total = workgroup sum
__device__ __forceinline__ u64 kc_block_excl_scan(u64 v, u64 *total) {
    __shared__ u64 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 inc = kc_wave_incl_scan(v);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (i < wv) base += wsum[i]; tot += wsum[i]; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}
// List space for a 256-thread workgroup with ONE global atomic: every thread calls it (no early exits before) with the number
// of items it wants to append; returns the position of its first item.  (One atomicAdd per item on a shared cursor serialises
// at ~12 ns per add on this GPU -- with 10^7 items that is the whole kernel.)
__device__ __forceinline__ u64 kc_block_reserve(u32 count, u64 *cursor) {
    __shared__ u64 kc_reserve_base;
    u64 tot;
    const u64 ex = kc_block_excl_scan((u64)count, &tot);
    if (threadIdx.x == 0) kc_reserve_base = tot ? atomicAdd((unsigned long long *)cursor, (unsigned long long)tot) : 0;
    __syncthreads();
    const u64 base = kc_reserve_base;
    __syncthreads();   // (the word may be rewritten by the next call)
    return base + ex;
}
#endif
