// Small device utilities shared by the translation units: exclusive scan, ordered compaction (wave ballot + prefix scan),
// and the message routing of the hash-sharded path (bin fixed-size items by destination rank, all-to-all, and the way back).
// Hand-written for 64-wide wavefronts; no library calls.
#include "kc_internal.h"
#include "kc_comm.h"
#include "kc_scan_dev.h"

// ---------------------------------------------------------------------------------------------------- exclusive scan
// Three launches per level: per-tile sums (2048 elements per 256-thread workgroup), scan of the tile sums (recursively),
// tile-local scan with the tile's offset.  out may alias in; out[n] receives the total.
#define KC_SCAN_TILE 2048
__global__ void __launch_bounds__(256) kc_scan_sums_kernel(const u64 *__restrict__ in, u64 n, u64 *sums) {
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE;
    u64 s = 0;
    for (u32 i = threadIdx.x; i < KC_SCAN_TILE; i += 256) if (base + i < n) s += in[base + i];
    u64 tot;
    (void)kc_block_excl_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) kc_scan_tile_kernel(const u64 *in, u64 n, const u64 *__restrict__ tile_off, u64 *out, u64 ntiles) {
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE + (u64)threadIdx.x * 8;
    u64 v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    u64 tot;
    u64 ex = kc_block_excl_scan(s, &tot) + (tile_off ? tile_off[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
    if (blockIdx.x == ntiles - 1 && threadIdx.x == 255) out[n] = ex;   // the last thread of the last tile ends at the total
}
int kcu_exclusive_scan(kc_ctx *ctx, const u64 *in, u64 *out, u64 n, bool sync) {
    if (n == 0) { KC_HIP(ctx, hipMemsetAsync(out, 0, 8, ctx->stream)); return KC_OK; }
    const u64 ntiles = (n + KC_SCAN_TILE - 1) / KC_SCAN_TILE;
    if (ntiles == 1) {
        KC_LAUNCH(kc_scan_tile_kernel, dim3(1), dim3(256), 0, ctx->stream, in, n, (const u64 *)nullptr, out, (u64)1);
        KC_HIP(ctx, hipGetLastError());
        return KC_OK;
    }
    u64 *sums = nullptr;
    KC_TRY(kc_alloc(ctx, &sums, ntiles + 1));
    // (on every way out, a launch that KC_LAUNCH refused included: sync = false hands the temporaries back at once -- whoever gets them
    // next works on the same stream, behind us)
    auto rel = kc_defer([&] { if (sync) (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, sums); });
    KC_LAUNCH(kc_scan_sums_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, in, n, sums);
    int rc = kcu_exclusive_scan(ctx, sums, sums, ntiles, sync);
    if (rc == KC_OK) {
        KC_LAUNCH(kc_scan_tile_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, in, n, (const u64 *)sums, out, ntiles);
        if (hipGetLastError() != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "scan launch failed");
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------- ordered compaction
// out = the elements of `in` whose flag byte is non-zero, in order (stable).  Flags -> per-tile counts (wave ballot +
// popcount), scan of the counts, then every tile writes its survivors at the tile offset + its in-tile rank.
__global__ void __launch_bounds__(256) kc_flag_counts_kernel(const u8 *__restrict__ flags, u64 n, u64 *counts) {
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE;
    u64 c = 0;
    for (u32 i = threadIdx.x; i < KC_SCAN_TILE; i += 256) if (base + i < n && flags[base + i]) ++c;
    u64 tot;
    (void)kc_block_excl_scan(c, &tot);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) kc_flag_write_kernel(const u64 *__restrict__ in, const u8 *__restrict__ flags, u64 n,
                                                            const u64 *__restrict__ tile_off, u64 *out) {
    // thread t owns the 8 consecutive elements base + 8t .. +7, so that ranks follow the element order
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE + (u64)threadIdx.x * 8;
    u64 c = 0;
    u32 m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (base + i < n && flags[base + i]) { m |= 1u << i; ++c; }
    u64 tot;
    u64 pos = tile_off[blockIdx.x] + kc_block_excl_scan(c, &tot);
#pragma unroll
    for (int i = 0; i < 8; ++i) if ((m >> i) & 1) out[pos++] = in[base + i];
}
// d_total (device, optional) receives the number of survivors
int kcu_select_flagged(kc_ctx *ctx, const u64 *in, const u8 *flags, u64 n, u64 *out, u64 *d_total, bool sync) {
    if (n == 0) { if (d_total) KC_HIP(ctx, hipMemsetAsync(d_total, 0, 8, ctx->stream)); return KC_OK; }
    const u64 ntiles = (n + KC_SCAN_TILE - 1) / KC_SCAN_TILE;
    u64 *counts = nullptr;
    KC_TRY(kc_alloc(ctx, &counts, ntiles + 1));
    auto rel = kc_defer([&] { if (sync) (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, counts); });   // (on every way out)
    KC_LAUNCH(kc_flag_counts_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, flags, n, counts);
    int rc = kcu_exclusive_scan(ctx, counts, counts, ntiles, sync);
    if (rc == KC_OK) {
        KC_LAUNCH(kc_flag_write_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, in, flags, n, (const u64 *)counts, out);
        if (hipGetLastError() != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "compaction launch failed");
        if (rc == KC_OK && d_total && hipMemcpyAsync(d_total, counts + ntiles, 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------- stable radix sort
// LSD radix sort of (key, value) pairs on the bits [begin_bit, end_bit) of 64-bit keys, 4 bits per pass: per-tile digit
// counts, one exclusive scan over (digit, tile), stable scatter (every thread owns 8 consecutive elements, its 16 counters
// live in registers, the cross-thread offsets come from workgroup prefix sums).  Used for the simplitig order, the hit
// lists of the overlap levels and the sparse branch; sizes are 10^3..10^7, far from the hot path.
__global__ void __launch_bounds__(256) kc_radix_hist_kernel(const u64 *__restrict__ keys, u64 n, int shift, u64 ntiles, u64 *counts) {
    __shared__ u32 h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE + (u64)threadIdx.x * 8;
    u32 c[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (base + i < n) {
            const u32 d = (u32)(keys[base + i] >> shift) & 15u;
#pragma unroll
            for (int b = 0; b < 16; ++b) c[b] += d == (u32)b;
        }
#pragma unroll
    for (int b = 0; b < 16; ++b) if (c[b]) atomicAdd(&h[b], c[b]);
    __syncthreads();
    if (threadIdx.x < 16) counts[(u64)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}
__global__ void __launch_bounds__(256) kc_radix_scatter_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ vals, u64 n, int shift, u64 ntiles,
                                                               const u64 *__restrict__ offsets, u64 *keys_out, u64 *vals_out) {
    const u64 base = (u64)blockIdx.x * KC_SCAN_TILE + (u64)threadIdx.x * 8;
    u64 k[8];
    u32 d[8], c[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d[i] = 16;
        if (base + i < n) {
            k[i] = keys[base + i];
            d[i] = (u32)(k[i] >> shift) & 15u;
#pragma unroll
            for (int b = 0; b < 16; ++b) c[b] += d[i] == (u32)b;
        }
    }
    u64 pos[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        u64 tot;
        pos[b] = offsets[(u64)b * ntiles + blockIdx.x] + kc_block_excl_scan((u64)c[b], &tot);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (d[i] == 16) continue;
        u64 p = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) if (d[i] == (u32)b) { p = pos[b]; pos[b] = p + 1; }
        keys_out[p] = k[i];
        if (vals) vals_out[p] = vals[base + i];
    }
}
// keys_in / vals_in are left intact; the sorted pairs end in keys_out / vals_out (vals may be null: keys only).
int kcu_sort_pairs(kc_ctx *ctx, const u64 *keys_in, u64 *keys_out, const u64 *vals_in, u64 *vals_out, u64 n, int begin_bit, int end_bit, bool sync) {
    if (n == 0) return KC_OK;
    const int passes = (end_bit - begin_bit + 3) / 4;
    if (passes <= 0) {
        KC_HIP(ctx, hipMemcpyAsync(keys_out, keys_in, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (vals_in) KC_HIP(ctx, hipMemcpyAsync(vals_out, vals_in, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return KC_OK;
    }
    const u64 ntiles = (n + KC_SCAN_TILE - 1) / KC_SCAN_TILE;
    u64 *counts = nullptr, *tk = nullptr, *tv = nullptr;
    KC_TRY(kc_alloc(ctx, &counts, 16 * ntiles + 1));
    int rc = kc_alloc(ctx, &tk, n);
    if (rc == KC_OK && vals_in) rc = kc_alloc(ctx, &tv, n);
    // ping-pong so that the last pass lands in the caller's output buffers
    const u64 *sk = keys_in, *sv = vals_in;
    for (int p = 0; p < passes && rc == KC_OK; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        u64 *dk = to_out ? keys_out : tk, *dv = to_out ? vals_out : tv;
        const int shift = begin_bit + 4 * p;
        KC_LAUNCH(kc_radix_hist_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, sk, n, shift, ntiles, counts);
        rc = kcu_exclusive_scan(ctx, counts, counts, 16 * ntiles, sync);
        if (rc != KC_OK) break;
        KC_LAUNCH(kc_radix_scatter_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, sk, sv, n, shift, ntiles, (const u64 *)counts, dk, dv);
        if (hipGetLastError() != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "radix sort launch failed");
        sk = dk; sv = dv;
    }
    if (sync) (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, counts); kc_release(ctx, tk); kc_release(ctx, tv);
    return rc;
}

// ---------------------------------------------------------------------------------------------------- bucketed row sort
// Rows = (W-word key with `total_bits` significant bits, id); wanted: ascending by (key, id).  One MSD pass on the top NB bits
// (histogram, single-workgroup scan, scatter) splits the rows into ~n/300 buckets, then one workgroup per bucket sorts its
// rows by (key, id) in LDS (bitonic network): four launches, where the LSD radix sort needs 5 per 4 bits of key.  A bucket
// beyond what a workgroup holds (skewed keys: repeats, low-complexity sequence) raises *d_flag and the output is void --
// the caller then falls back to the LSD sort.  ids must be unique (they break ties).
#define KC_BKT_THREADS 256
__device__ __forceinline__ u32 kc_row_bucket(const u64 *__restrict__ row, int W, int total_bits, int nb_bits) {
    const int lo = total_bits - nb_bits, w0 = lo >> 6, off = lo & 63;
    u64 v = row[w0] >> off;
    if (off + nb_bits > 64 && w0 + 1 < W) v |= row[w0 + 1] << (64 - off);
    return (u32)v & ((1u << nb_bits) - 1u);
}
// (cstride: with few buckets every counter sits on a line of its own -- atomics on one 128-byte line serialise at ~4 ns each,
// which made the two atomic passes of a 16 000-row sort cost more than the sort)
__global__ void __launch_bounds__(256) kc_bkt_hist_kernel(const u64 *__restrict__ keys, u64 n, int W, int total_bits, int nb_bits, u32 *counts, u32 cstride) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[(size_t)kc_row_bucket(keys + i * (u64)W, W, total_bits, nb_bits) * cstride], 1u);
}
// one workgroup: offsets[b] = rows before bucket b (offsets[nb] = n), cursors = a copy to scatter through; *flag |= a bucket exceeds cap
__global__ void __launch_bounds__(1024) kc_bkt_scan_kernel(const u32 *__restrict__ counts, u32 nb, u32 cap, u32 *offsets, u32 *cursors, u64 *flag, u32 cstride) {
    __shared__ u32 wsum[16];
    __shared__ u32 over;
    const u32 per = (nb + 1023u) / 1024u, lo = threadIdx.x * per;
    if (threadIdx.x == 0) over = 0;
    __syncthreads();
    u32 s = 0, big = 0;
    for (u32 i = lo; i < lo + per && i < nb; ++i) { const u32 c = counts[(size_t)i * cstride]; s += c; big |= c > cap; }
    if (big) over = 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    u32 base = 0;
    for (int i = 0; i < wv; ++i) base += wsum[i];
    u32 run = base + inc - s;
    for (u32 i = lo; i < lo + per && i < nb; ++i) { offsets[i] = run; cursors[(size_t)i * cstride] = run; run += counts[(size_t)i * cstride]; }
    if (threadIdx.x == 1023) offsets[nb] = run;
    if (threadIdx.x == 0 && over) *flag = 1;
}
__global__ void __launch_bounds__(256) kc_bkt_scatter_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ ids, u64 n, int W, int total_bits,
                                                             int nb_bits, u32 *cursors, u64 *tk, u64 *tv, u32 cstride) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 *row = keys + i * (u64)W;
    const u32 pos = atomicAdd(&cursors[(size_t)kc_row_bucket(row, W, total_bits, nb_bits) * cstride], 1u);
    for (int j = 0; j < W; ++j) tk[(u64)pos * W + j] = row[j];
    tv[pos] = ids[i];
}
__global__ void __launch_bounds__(KC_BKT_THREADS) kc_bkt_sort_kernel(const u64 *__restrict__ tk, const u64 *__restrict__ tv, const u32 *__restrict__ offsets,
                                                                     int W, u32 cap, const u64 *__restrict__ flag, u64 *keys_out, u64 *ids_out) {
    extern __shared__ u64 kc_bkt[];   // cap ids, then W planes of cap key words
    const u32 lo = offsets[blockIdx.x], m = offsets[blockIdx.x + 1] - lo;
    if (m == 0 || *flag) return;
    if (m == 1) {
        if (threadIdx.x == 0) { ids_out[lo] = tv[lo]; for (int j = 0; j < W; ++j) keys_out[(u64)lo * W + j] = tk[(u64)lo * W + j]; }
        return;
    }
    u32 P = 2;
    while (P < m) P <<= 1;
    u64 *sid = kc_bkt, *sk = kc_bkt + cap;
    for (u32 i = threadIdx.x; i < P; i += blockDim.x) {
        sid[i] = i < m ? tv[lo + i] : ~(u64)0;
        for (int j = 0; j < W; ++j) sk[(u32)j * cap + i] = i < m ? tk[(u64)(lo + i) * W + j] : ~(u64)0;
    }
    __syncthreads();
    auto greater = [&](u32 a, u32 b) -> bool {   // row a sorts after row b
        for (int j = W - 1; j >= 0; --j) {
            const u64 x = sk[(u32)j * cap + a], y = sk[(u32)j * cap + b];
            if (x != y) return x > y;
        }
        return sid[a] > sid[b];
    };
    for (u32 size = 2; size <= P; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            for (u32 t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const u32 a = ((t / stride) * (stride << 1)) + (t % stride), b = a + stride;
                const bool up = ((a & size) == 0);
                if (greater(a, b) == up) {
                    u64 tmp = sid[a]; sid[a] = sid[b]; sid[b] = tmp;
                    for (int j = 0; j < W; ++j) { tmp = sk[(u32)j * cap + a]; sk[(u32)j * cap + a] = sk[(u32)j * cap + b]; sk[(u32)j * cap + b] = tmp; }
                }
            }
            __syncthreads();
        }
    }
    for (u32 i = threadIdx.x; i < m; i += blockDim.x) {
        ids_out[lo + i] = sid[i];
        for (int j = 0; j < W; ++j) keys_out[(u64)(lo + i) * W + j] = sk[(u32)j * cap + i];
    }
}
// returns KC_E_ARG when the shape does not suit the method (the caller then uses the LSD sort right away); otherwise the sort
// is enqueued and *d_flag (device word, zeroed here) tells afterwards whether a bucket overflowed.  No stream sync.
int kcu_sort_rows_bucketed(kc_ctx *ctx, const u64 *keys, const u64 *ids, u64 n, int W, int total_bits, u64 *keys_out, u64 *ids_out, u64 *d_flag) {
    if (n < 1024 || n >= ((u64)1 << 31)) return KC_E_ARG;
    int lg = 0;
    while (((u64)1 << lg) < n) ++lg;
    int nb_bits = lg - 8;   // ~256-512 rows per bucket when the keys are spread evenly
    if (nb_bits < 3) nb_bits = 3;
    if (nb_bits > 16) nb_bits = 16;
    if (total_bits < nb_bits + 4) return KC_E_ARG;   // (few key bits: the LSD sort needs few passes anyway)
    const u32 nb = 1u << nb_bits;
    const u32 cap = W == 4 ? 1024u : 2048u;
    const size_t lds = (size_t)cap * (size_t)(W + 1) * 8;
    u32 *counts = nullptr, *offsets = nullptr, *cursors = nullptr;
    u64 *tk = nullptr, *tv = nullptr;
    const u32 cstride = nb <= 2048 ? 32u : 1u;
    int rc = kc_alloc(ctx, &counts, (size_t)nb * cstride);
    if (rc == KC_OK) rc = kc_alloc(ctx, &offsets, nb + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &cursors, (size_t)nb * cstride);
    if (rc == KC_OK) rc = kc_alloc(ctx, &tk, n * (u64)W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &tv, n);
    // (the temporaries go back to the pool without a stream sync: their next user queues behind these kernels)
    auto done = [&](int r) { kc_release(ctx, counts); kc_release(ctx, offsets); kc_release(ctx, cursors); kc_release(ctx, tk); kc_release(ctx, tv); return r; };
    if (rc != KC_OK) return done(rc);
    if (hipMemsetAsync(counts, 0, (size_t)nb * cstride * 4, ctx->stream) != hipSuccess || hipMemsetAsync(d_flag, 0, 8, ctx->stream) != hipSuccess)
        return done(kc_fail(ctx, KC_E_NO_DEVICE, "memset failed"));
    KC_LAUNCH(kc_bkt_hist_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, keys, n, W, total_bits, nb_bits, counts, cstride);
    KC_LAUNCH(kc_bkt_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const u32 *)counts, nb, cap, offsets, cursors, d_flag, cstride);
    KC_LAUNCH(kc_bkt_scatter_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, keys, ids, n, W, total_bits, nb_bits, cursors, tk, tv, cstride);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kc_bkt_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return done(kc_fail(ctx, KC_E_NO_DEVICE, "LDS size refused"));
    KC_LAUNCH(kc_bkt_sort_kernel, dim3(nb), dim3(KC_BKT_THREADS), lds, ctx->stream, (const u64 *)tk, (const u64 *)tv, (const u32 *)offsets, W, cap,
                       (const u64 *)d_flag, keys_out, ids_out);
    if (hipGetLastError() != hipSuccess) return done(kc_fail(ctx, KC_E_NO_DEVICE, "bucketed sort launch failed"));
    return done(KC_OK);
}

// ---------------------------------------------------------------------------------------------------- sharded lists
// Lists filled through KC_USHARDS cursors (one 128-byte line each): shard s owns the items [s * shard_cap, s * shard_cap + min(count_s, shard_cap)).
// kcu_compact_shards copies them, E words per item, into one contiguous list and returns the total (or ~0 on overflow of a shard).
__global__ void __launch_bounds__(256) kc_compact_shards_kernel(const u64 *__restrict__ src, u64 shard_cap, const u64 *__restrict__ cursors, int nshards,
                                                                int stride, int E, u64 *dst) {
    __shared__ u64 pre;
    const int sh = blockIdx.y;
    if (threadIdx.x == 0) {
        u64 a = 0;
        for (int s = 0; s < sh; ++s) { u64 c = cursors[(size_t)s * stride]; a += c > shard_cap ? shard_cap : c; }
        pre = a;
    }
    __syncthreads();
    u64 n = cursors[(size_t)sh * stride];
    if (n > shard_cap) n = shard_cap;
    const u64 words = n * (u64)E;
    const u64 *s0 = src + (u64)sh * shard_cap * (u64)E;
    u64 *d0 = dst + pre * (u64)E;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (u64)gridDim.x * blockDim.x) d0[i] = s0[i];
}
int kcu_compact_shards(kc_ctx *ctx, const u64 *src, u64 shard_cap, const u64 *d_cursors, int nshards, int stride, int E, u64 *dst, u64 *total, bool *overflow) {
    std::vector<u64> h((size_t)nshards * stride);
    KC_HIP(ctx, hipMemcpyAsync(h.data(), d_cursors, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 t = 0;
    bool ovf = false;
    for (int s = 0; s < nshards; ++s) { u64 c = h[(size_t)s * stride]; if (c > shard_cap) { ovf = true; c = shard_cap; } t += c; }
    *total = t; *overflow = ovf;
    if (t == 0 || ovf) return KC_OK;
    KC_LAUNCH(kc_compact_shards_kernel, dim3(64, nshards), dim3(256), 0, ctx->stream, src, shard_cap, d_cursors, nshards, stride, E, dst);
    KC_HIP(ctx, hipGetLastError());
    return KC_OK;
}

// ---------------------------------------------------------------------------------------------------- routing
// n items, item i goes to rank dest[i].  kcr_build bins them (LDS histogram per workgroup, one reservation per
// destination and workgroup) into a permutation `perm` (send position -> item) and exchanges the counts;
// kcr_forward moves E-word items to their destinations, kcr_backward brings A-word answers back into item order.
#define KC_MAX_RANKS 64
__global__ void __launch_bounds__(256) kc_route_count_kernel(const u8 *__restrict__ dest, u64 n, u64 *counts) {
    __shared__ u32 h[KC_MAX_RANKS];
    if (threadIdx.x < KC_MAX_RANKS) h[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&h[dest[i]], 1u);
    __syncthreads();
    if (threadIdx.x < KC_MAX_RANKS && h[threadIdx.x]) atomicAdd((ull *)&counts[threadIdx.x], (ull)h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) kc_route_perm_kernel(const u8 *__restrict__ dest, u64 n, const u64 *__restrict__ off, u64 *cursors, u64 *perm) {
    __shared__ u32 h[KC_MAX_RANKS];
    __shared__ u64 base[KC_MAX_RANKS];
    const u64 tile = (u64)blockIdx.x * 2048;
    if (threadIdx.x < KC_MAX_RANKS) h[threadIdx.x] = 0;
    __syncthreads();
    u32 mine[8]; u8 d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 i = tile + (u64)j * 256 + threadIdx.x;
        d[j] = 0; mine[j] = 0;
        if (i < n) { d[j] = dest[i]; mine[j] = atomicAdd(&h[d[j]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < KC_MAX_RANKS && h[threadIdx.x]) base[threadIdx.x] = off[threadIdx.x] + atomicAdd((ull *)&cursors[threadIdx.x], (ull)h[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 i = tile + (u64)j * 256 + threadIdx.x;
        if (i < n) perm[base[d[j]] + mine[j]] = i;
    }
}
__global__ void __launch_bounds__(256) kc_route_gather_kernel(const u64 *__restrict__ items, const u64 *__restrict__ perm, u64 n, int E, int stride, u64 *send) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 *s = items + perm[t] * (u64)stride;
    for (int w = 0; w < E; ++w) send[t * (u64)E + w] = s[w];
}
__global__ void __launch_bounds__(256) kc_route_scatter_kernel(const u64 *__restrict__ tmp, const u64 *__restrict__ perm, u64 n, int A, u64 *out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 *d = out + perm[t] * (u64)A;
    for (int w = 0; w < A; ++w) d[w] = tmp[t * (u64)A + w];
}

int kcr_build(kc_comm *c, const u8 *dest, u64 n, KcRoute *r) {
    kc_ctx *ctx = c->ctx;
    const int G = c->world;
    r->n = n; r->perm = nullptr;
    r->scnt.assign(G, 0); r->soff.assign(G, 0); r->rcnt.assign(G, 0); r->roff.assign(G, 0);
    u64 *cnt = nullptr;
    KC_TRY(kc_alloc(ctx, &cnt, 3 * KC_MAX_RANKS));
    KC_HIP(ctx, hipMemsetAsync(cnt, 0, 3 * KC_MAX_RANKS * 8, ctx->stream));
    if (n) {
        unsigned g = kc_grid(n, 256 * 16);
        KC_LAUNCH(kc_route_count_kernel, dim3(g), dim3(256), 0, ctx->stream, dest, n, cnt);
    }
    u64 h[KC_MAX_RANKS];
    KC_HIP(ctx, hipMemcpyAsync(h, cnt, (size_t)G * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 o = 0;
    for (int p = 0; p < G; ++p) { r->scnt[p] = h[p]; r->soff[p] = o; o += h[p]; }
    // every rank learns what it will receive from whom
    std::vector<u64> all((size_t)G * G);
    int rc = kcc_allgather_words(c, r->scnt.data(), all.data(), G);
    if (rc != KC_OK) { kc_release(ctx, cnt); return rc; }
    o = 0;
    for (int p = 0; p < G; ++p) { r->rcnt[p] = all[(size_t)p * G + c->rank]; r->roff[p] = o; o += r->rcnt[p]; }
    r->n_recv = o;
    r->n_total = 0;
    for (u64 v : all) r->n_total += v;
    if (n) {
        rc = kc_alloc(ctx, &r->perm, n);
        if (rc != KC_OK) { kc_release(ctx, cnt); return rc; }
        KC_HIP(ctx, hipMemcpyAsync(cnt + KC_MAX_RANKS, r->soff.data(), (size_t)G * 8, hipMemcpyHostToDevice, ctx->stream));
        KC_LAUNCH(kc_route_perm_kernel, dim3(kc_grid(n, 2048)), dim3(256), 0, ctx->stream, dest, n, (const u64 *)(cnt + KC_MAX_RANKS), cnt + 2 * KC_MAX_RANKS, r->perm);
        KC_HIP(ctx, hipGetLastError());
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (the offsets were copied from a host vector)
    }
    kc_release(ctx, cnt);
    return KC_OK;
}
void kcr_free(kc_ctx *ctx, KcRoute *r) { if (r->perm) { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, r->perm); r->perm = nullptr; } }

// items: n rows of `stride` words, the first E of which travel.  *recv_out (n_recv * E words) is allocated here.
int kcr_forward(kc_comm *c, const KcRoute &r, const u64 *items, int E, int stride, u64 **recv_out) {
    kc_ctx *ctx = c->ctx;
    const int G = c->world;
    u64 *send = nullptr, *recv = nullptr;
    KC_TRY(kc_alloc(ctx, &send, r.n * (u64)E + 1));
    int rc = kc_alloc(ctx, &recv, r.n_recv * (u64)E + 1);
    if (rc != KC_OK) { kc_release(ctx, send); return rc; }
    bool keep_recv = false;
    auto rel = kc_defer([&] { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, send); if (!keep_recv) kc_release(ctx, recv); });   // (a refused launch included)
    if (r.n) KC_LAUNCH(kc_route_gather_kernel, dim3(kc_grid(r.n, 256)), dim3(256), 0, ctx->stream, items, (const u64 *)r.perm, r.n, E, stride, send);
    std::vector<u64> so(G), sc(G), ro(G), rcv(G);
    for (int p = 0; p < G; ++p) { so[p] = r.soff[p] * E * 8; sc[p] = r.scnt[p] * E * 8; ro[p] = r.roff[p] * E * 8; rcv[p] = r.rcnt[p] * E * 8; }
    rc = kcc_alltoallv(c, send, so.data(), sc.data(), recv, ro.data(), rcv.data());
    if (rc != KC_OK) return rc;
    keep_recv = true;
    *recv_out = recv;
    return KC_OK;
}
// answers: n_recv rows of A words (in the order the requests arrived); out: n rows of A words in item order
int kcr_backward(kc_comm *c, const KcRoute &r, const u64 *answers, int A, u64 *out) {
    kc_ctx *ctx = c->ctx;
    const int G = c->world;
    u64 *tmp = nullptr;
    KC_TRY(kc_alloc(ctx, &tmp, r.n * (u64)A + 1));
    auto rel = kc_defer([&] { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, tmp); });   // (a refused launch included)
    std::vector<u64> so(G), sc(G), ro(G), rcv(G);
    for (int p = 0; p < G; ++p) { so[p] = r.roff[p] * A * 8; sc[p] = r.rcnt[p] * A * 8; ro[p] = r.soff[p] * A * 8; rcv[p] = r.scnt[p] * A * 8; }
    int rc = kcc_alltoallv(c, answers, so.data(), sc.data(), tmp, ro.data(), rcv.data());
    if (rc == KC_OK && r.n) {
        KC_LAUNCH(kc_route_scatter_kernel, dim3(kc_grid(r.n, 256)), dim3(256), 0, ctx->stream, (const u64 *)tmp, (const u64 *)r.perm, r.n, A, out);
        if (hipGetLastError() != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "route scatter launch failed");
    }
    return rc;
}
