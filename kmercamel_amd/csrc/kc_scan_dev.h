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
#endif
