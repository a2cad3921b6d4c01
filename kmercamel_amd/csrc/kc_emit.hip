// Phase 3 of the BIGREEDY path on gfx950: masked superstring from the overlap path.
// Replaces SuperstringFromPath / SuperstringFromPathSparse (global.h:136-210, global_sparse.h:138-202) and, as the
// first "next" row, OptimizeOnes (masks.h:40-78).
//
// The reference walks edgeFrom sequentially.  Here: the start vertex is the smallest vertex without in-edge
// (global.h:156-161); every vertex v on its path contributes the block simplitig_v[0 .. len_v - ov_v) (upper case up
// to index len_v - k, lower case after; ov_v = overlap of the edge leaving v, 0 for the last vertex, which yields the
// trailing k-1 lower-case characters of global.h:202-205); block offsets come from a weighted list ranking by pointer
// jumping; one wavefront per vertex then writes its block with coalesced byte stores.  The -M mask (global.h:185-195)
// is a per-position probe of the k-mer set over the finished string.
#include "kc_internal.h"

__global__ void __launch_bounds__(256) em_pred_kernel(const u64 *__restrict__ edge_from, u64 V, u64 *pred) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    u64 e = edge_from[v];
    if (e != KC_NONE64) pred[e] = v;
}
__global__ void __launch_bounds__(256) em_start_kernel(const u64 *__restrict__ pred, u64 V, u64 *start) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    if (pred[v] == KC_NONE64) atomicMin((ull *)start, (ull)v);
}
__global__ void __launch_bounds__(256) em_iota_kernel(u64 *a, u64 n) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = t;
}
// pointer-jumping state of a vertex in ONE 32-byte record: a jump reads the successor's record with a single scattered access
// (three separate arrays cost three lines per vertex and round: 265 -> ~100 us per round at V = 5e6)
struct __attribute__((aligned(32))) EmRank { u64 nxt, dist, endv, pad; };
__global__ void __launch_bounds__(256) em_rank_init_kernel(const u64 *__restrict__ edge_from, const u8 *__restrict__ overlaps,
                                                           const u64 *__restrict__ lens, u64 S, u64 V, EmRank *st, u64 *weight) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    u64 e = edge_from[v];
    u64 L = lens[v < S ? v : v - S];
    u64 w = L - (e != KC_NONE64 ? (u64)overlaps[v] : 0);
    st[v] = EmRank{e, w, v, 0};
    weight[v] = w;
}
__global__ void __launch_bounds__(256) em_rank_step_kernel(const EmRank *__restrict__ a, u64 V, EmRank *b) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    EmRank x = a[v];
    if (x.nxt != KC_NONE64) { const EmRank y = a[x.nxt]; x.nxt = y.nxt; x.dist += y.dist; x.endv = y.endv; }
    b[v] = x;
}
__global__ void __launch_bounds__(256) em_rank_split_kernel(const EmRank *__restrict__ a, u64 V, u64 *dist, u64 *endv) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) { dist[v] = a[v].dist; endv[v] = a[v].endv; }
}
// ---- sampled list ranking (large V).  Pointer jumping over all V vertices takes log2 V rounds of V scattered records; instead
// every ~32nd vertex (by a hash of its index) and every path head is a SPLITTER: one thread per splitter walks to the next
// splitter, summing weights and telling every vertex it passes its splitter and the weight before it; pointer jumping then
// runs over the splitters only, and a last pass gives every vertex dist = dist(splitter) - weight before it.  Same numbers.
#define KC_RANK_SAMPLE_MIN ((u64)1 << 20)   /* below this the plain rounds are cheaper (measured: 2e5 vertices 0.42 vs 0.49 ms) */
__device__ __forceinline__ bool em_is_sampled(u64 v) { return ((v * 0x9E3779B97F4A7C15ULL) >> 59) == 0; }   // 1 in 32
__global__ void __launch_bounds__(256) em_mark_splitters_kernel(EmRank *st, const u64 *__restrict__ pred, u64 V, u8 *flag) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const bool sp = pred[v] == KC_NONE64 || em_is_sampled(v);
    st[v].pad = sp ? 1 : 0;
    flag[v] = sp ? 1 : 0;
}
__global__ void __launch_bounds__(256) em_segment_walk_kernel(const EmRank *__restrict__ st, const u64 *__restrict__ spl, u64 ns, EmRank *seg, u32 *own, u64 *before) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns) return;
    const u64 x = spl[t];
    u64 cur = x, acc = 0;
    while (true) {
        const EmRank r = st[cur];
        if (cur != x && r.pad) { seg[x] = EmRank{cur, acc, x, 1}; return; }   // the next splitter ends this segment
        own[cur] = (u32)x; before[cur] = acc;
        acc += r.dist;
        if (r.nxt == KC_NONE64) { seg[x] = EmRank{KC_NONE64, acc, cur, 1}; return; }   // the path ends inside this segment
        cur = r.nxt;
    }
}
__global__ void __launch_bounds__(256) em_splitter_step_kernel(const EmRank *__restrict__ a, const u64 *__restrict__ spl, u64 ns, EmRank *b) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns) return;
    const u64 x = spl[t];
    EmRank r = a[x];
    if (r.nxt != KC_NONE64) { const EmRank y = a[r.nxt]; r.nxt = y.nxt; r.dist += y.dist; r.endv = y.endv; }
    b[x] = r;
}
__global__ void __launch_bounds__(256) em_rank_finish_kernel(const EmRank *__restrict__ seg, const u32 *__restrict__ own, const u64 *__restrict__ before, u64 V,
                                                             u64 *dist, u64 *endv) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const EmRank r = seg[own[v]];
    dist[v] = r.dist - before[v];
    endv[v] = r.endv;
}

// G lanes per vertex on the start path: a wavefront (64) for simplitigs of hundreds of characters and more, 8 lanes for short ones
// (reads with errors: 2 10^7 simplitigs of ~48 characters -- a wavefront each left 62 lanes idle behind six dependent loads)
template <int G>
__global__ void __launch_bounds__(256) em_emit_kernel(const u64 *__restrict__ words, const u64 *__restrict__ woff,
                                                      const u64 *__restrict__ lens, u64 S, u64 V, int k, const u64 *__restrict__ dist,
                                                      const u64 *__restrict__ endv, const u64 *__restrict__ weight, u64 start, char *out,
                                                      u64 out_lo, int nranks, int me) {
    // nranks > 1 (hash-sharded flow): this rank emits the vertices whose block starts in its share of the output, into a
    // buffer that begins at output offset out_lo
    constexpr int LG = G == 64 ? 6 : G == 16 ? 4 : 3;
    static_assert(G == 64 || G == 16 || G == 8, "lanes per vertex");
    u64 v = (((u64)blockIdx.z * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x) >> LG;   // (x, z): vertices; y: slices of long vertices
    int lane = threadIdx.x & (G - 1);
    if (v >= V) return;
    if (endv[v] != endv[start]) return;
    u64 off = dist[start] - dist[v];
    if (nranks > 1 && (int)(((unsigned __int128)off * (unsigned)nranks) / dist[start]) != me) return;
    off -= out_lo;
    u64 sidx = v < S ? v : v - S;
    bool rc = v >= S;
    u64 L = lens[sidx], wb = woff[sidx], n = weight[v];
    u64 upper_end = L - (u64)k;   // indices <= upper_end are k-mer starts of the simplitig
    // every lane spells 32 consecutive characters per step -- one packed word forwards; backwards the 32 bases that end at L-1-t0,
    // complemented and reversed as a word -- in four 8-byte stores; blockIdx.y splits long vertices (assembled genomes give
    // simplitigs of Mbp) into interleaved 2048-character slices
    for (u64 t0 = ((u64)blockIdx.y * G + lane) * 32; t0 < n; t0 += (u64)gridDim.y * (G * 32)) {
        const int cnt = (n - t0 >= 32) ? 32 : (int)(n - t0);
        u64 w;   // the 32 bases in the order they are written, first one in the highest pair
        if (!rc) w = words[wb + (t0 >> 5)];
        else {
            const u64 jh = L - 1 - t0;   // base of the first character; the window is [jh - 31, jh]
            u64 X;
            if (jh < 31) X = words[wb] >> (2 * (31 - jh));
            else {
                const u64 ws = jh - 31;
                const int sh = (int)(ws & 31) * 2;
                const u64 wa = words[wb + (ws >> 5)];
                X = sh ? ((wa << sh) | (words[wb + (ws >> 5) + 1] >> (64 - sh))) : wa;
            }
            const u64 r = __brevll(~X);   // pairs reversed, but the two bits of each pair swapped: swap them back
            w = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (8 * g >= cnt) break;
            u64 packed = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u64 bse = (w >> (62 - 2 * (8 * g + i))) & 3;
                u64 c = (0x54474341ull >> (8 * bse)) & 0xffull;   // "ACGT"
                if (t0 + (u64)(8 * g + i) > upper_end) c += (u64)('a' - 'A');
                packed |= c << (8 * i);
            }
            const int m = cnt - 8 * g;
            if (m >= 8) __builtin_memcpy(out + off + t0 + 8 * g, &packed, 8);
            else for (int i = 0; i < m; ++i) out[off + t0 + (u64)(8 * g + i)] = (char)(packed >> (8 * i));
        }
    }
}

// -M: every lower-case position p <= L-k is switched on iff the k-mer starting at p is in the set (global.h:185-195)
template <int W>
__global__ void __launch_bounds__(256) em_mask_kernel(const char *__restrict__ out, u64 L, int k, int complements, SetView<W> set, char *mask) {
    u64 p = kc_gtid();
    if (p >= L) return;
    char c = out[p];
    if (c <= 'Z' || p + (u64)k > L) { mask[p] = c; return; }
    Key<W> x = key_zero<W>();
    for (int i = 0; i < k; ++i) {
        u32 up = ((u32)(u8)out[p + i]) & 0xDFu;
        u32 code = (up >> 1) & 3u;
        code ^= code >> 1;
        x = key_append(x, (u64)code, k);
    }
    if (complements) { Key<W> r = key_rc(x, k); if (key_lt(r, x)) x = r; }
    bool in = set_find(set, x) != KC_NONE64;
    mask[p] = in ? (char)(c - ('a' - 'A')) : c;
}

// weighted list ranking of the path: dist[v] = characters from v's block to the end of its path, endv[v] = last vertex of
// v's path, weight[v] = characters v contributes; start = smallest vertex without in-edge; L = dist[start]
int kce_rank_path(kc_simplitigs *s, const KcPathDev *path, KcRank *r) {
    kc_ctx *ctx = s->ctx;
    KcScope scope(ctx);
    const u64 S = s->n, V = path->V;
    u64 *pred = nullptr, *start = nullptr, *dist, *endv, *weight;
    EmRank *ra = nullptr, *rb = nullptr;
    KC_TRY(scope.alloc(&pred, V)); KC_TRY(scope.alloc(&start, 2));
    KC_TRY(scope.alloc(&ra, V + 1)); KC_TRY(scope.alloc(&rb, V + 1));
    KC_TRY(scope.alloc(&dist, V)); KC_TRY(scope.alloc(&endv, V)); KC_TRY(scope.alloc(&weight, V));
    KC_HIP(ctx, hipMemsetAsync(pred, 0xff, V * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(start, 0xff, 16, ctx->stream));
    unsigned g = kc_grid(V, 256);
    KC_LAUNCH(em_pred_kernel, dim3(g), dim3(256), 0, ctx->stream, (const u64 *)path->edge_from, V, pred);
    KC_LAUNCH(em_start_kernel, dim3(g), dim3(256), 0, ctx->stream, (const u64 *)pred, V, start);
    KC_LAUNCH(em_rank_init_kernel, dim3(g), dim3(256), 0, ctx->stream, (const u64 *)path->edge_from, (const u8 *)path->overlaps,
                       (const u64 *)s->lens, S, V, ra, weight);
    u64 sample_min = KC_RANK_SAMPLE_MIN;
    if (const char *e = getenv("KC_SAMPLED_RANK_MIN")) { const long long v = atoll(e); if (v >= 0) sample_min = (u64)v; }   // (tests force the sampled route with 0)
    if (V < sample_min || V < 2 || V >= ((u64)1 << 32)) {
        for (u64 span = 1; span < V; span <<= 1) {
            KC_LAUNCH(em_rank_step_kernel, dim3(g), dim3(256), 0, ctx->stream, (const EmRank *)ra, V, rb);
            std::swap(ra, rb);
        }
        KC_LAUNCH(em_rank_split_kernel, dim3(g), dim3(256), 0, ctx->stream, (const EmRank *)ra, V, dist, endv);
    } else {
        u8 *flag = nullptr;
        u64 *iota = nullptr, *spl = nullptr, *before = nullptr, *d_ns = nullptr, ns = 0;
        u32 *own = nullptr;
        KC_TRY(scope.alloc(&flag, V)); KC_TRY(scope.alloc(&spl, V)); KC_TRY(scope.alloc(&before, V)); KC_TRY(scope.alloc(&own, V)); KC_TRY(scope.alloc(&d_ns, 2));
        KC_HIP(ctx, hipMemsetAsync(own, 0, V * 4, ctx->stream));   // (a vertex no walk reaches -- there is none on a path cover -- then points at record 0)
        iota = endv;   // (endv is written last: until then it holds 0..V-1 for the compaction)
        KC_LAUNCH(em_iota_kernel, dim3(g), dim3(256), 0, ctx->stream, iota, V);
        KC_LAUNCH(em_mark_splitters_kernel, dim3(g), dim3(256), 0, ctx->stream, ra, (const u64 *)pred, V, flag);
        KC_TRY(kcu_select_flagged(ctx, iota, flag, V, spl, d_ns, false));
        KC_HIP(ctx, hipMemcpyAsync(&ns, d_ns, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ns) {
            const unsigned gs = kc_grid(ns, 256);
            // segment records live at the splitters' own indices of rb / ra (the jumping rounds ping-pong between the two)
            KC_LAUNCH(em_segment_walk_kernel, dim3(gs), dim3(256), 0, ctx->stream, (const EmRank *)ra, (const u64 *)spl, ns, rb, own, before);
            EmRank *sa = rb, *sb = ra;   // (ra's vertex records are no longer needed once the walks are through)
            for (u64 span = 1; span < ns; span <<= 1) {
                KC_LAUNCH(em_splitter_step_kernel, dim3(gs), dim3(256), 0, ctx->stream, (const EmRank *)sa, (const u64 *)spl, ns, sb);
                std::swap(sa, sb);
            }
            KC_LAUNCH(em_rank_finish_kernel, dim3(g), dim3(256), 0, ctx->stream, (const EmRank *)sa, (const u32 *)own, (const u64 *)before, V, dist, endv);
        }
    }
    KC_HIP(ctx, hipGetLastError());
    u64 h_start = 0;
    KC_HIP(ctx, hipMemcpyAsync(&h_start, start, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_start == KC_NONE64) return kc_fail(ctx, KC_E_ARG, "overlap path has no start vertex (cycle cover?)");
    u64 L = 0;
    KC_HIP(ctx, hipMemcpyAsync(&L, dist + h_start, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    scope.keep(dist); scope.keep(endv); scope.keep(weight);
    r->V = V; r->dist = dist; r->endv = endv; r->weight = weight; r->start = h_start; r->L = L;
    return KC_OK;
}
void kce_rank_release(kc_ctx *ctx, KcRank *r) {
    kc_release(ctx, r->dist); kc_release(ctx, r->endv); kc_release(ctx, r->weight);
    r->dist = r->endv = r->weight = nullptr;
}
// spell the blocks of the start path into `out` (which begins at output offset out_lo; nranks > 1: only this rank's blocks)
int kce_emit_ranked(kc_simplitigs *s, const KcRank *r, u64 out_lo, int nranks, int me, char *out) {
    kc_ctx *ctx = s->ctx;
    const u64 V = r->V, L = r->L;
    unsigned slices = 1;   // waves per vertex: ~8192 characters each on average, at most 256
    if (V && L / V > 16384) { u64 y = L / V / 8192; slices = (unsigned)(y > 256 ? 256 : y); }
    // one wavefront per vertex: from 2^26 vertices on the block count no longer fits one grid dimension's 2^32 work-items
    bool short_ones = V && L / V < 256;   // (L / V: half the mean block, the mirror vertices are not on the path)
    // (in between -- reads of a kilobase: 16 lanes, so that a wavefront has four vertices' dependent loads in flight instead of one's)
    bool mid_ones = V && !short_ones && L / V < 2048;
    if (const char *e = getenv("KC_EMIT_LANES")) { short_ones = atoi(e) == 8; mid_ones = atoi(e) == 16; }   // (test knob: 8, 16 or 64)
    const u64 lanes = short_ones ? 8 : mid_ones ? 16 : 64;
    const u64 gv = (V * lanes + 255) / 256 ? (V * lanes + 255) / 256 : 1;
    const unsigned gx = gv > ((u64)1 << 22) ? (1u << 22) : (unsigned)gv, gz = (unsigned)((gv + gx - 1) / gx);
    if (short_ones)
        KC_LAUNCH(em_emit_kernel<8>, dim3(gx, slices, gz), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff,
                           (const u64 *)s->lens, s->n, V, s->k, (const u64 *)r->dist, (const u64 *)r->endv, (const u64 *)r->weight, r->start, out, out_lo, nranks, me);
    else if (mid_ones)
        KC_LAUNCH(em_emit_kernel<16>, dim3(gx, slices, gz), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff,
                           (const u64 *)s->lens, s->n, V, s->k, (const u64 *)r->dist, (const u64 *)r->endv, (const u64 *)r->weight, r->start, out, out_lo, nranks, me);
    else
        KC_LAUNCH(em_emit_kernel<64>, dim3(gx, slices, gz), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff,
                           (const u64 *)s->lens, s->n, V, s->k, (const u64 *)r->dist, (const u64 *)r->endv, (const u64 *)r->weight, r->start, out, out_lo, nranks, me);
    KC_HIP(ctx, hipGetLastError());
    return KC_OK;
}

int kce_superstring(kc_simplitigs *s, const KcPathDev *path, int complements, kc_set *set_for_mask, char **d_out, char **d_mask, u64 *length) {
    kc_ctx *ctx = s->ctx;
    const int k = s->k;
    KcRank rk;
    KC_TRY(kce_rank_path(s, path, &rk));
    const u64 L = rk.L;
    char *out = nullptr, *mask = nullptr;
    int rc = kc_alloc(ctx, &out, L + 1);
    if (rc == KC_OK) rc = kce_emit_ranked(s, &rk, 0, 1, 0, out);
    if (rc == KC_OK && d_mask && set_for_mask) {
        rc = kc_alloc(ctx, &mask, L + 1);
        if (rc == KC_OK) {
            const dim3 gm = kc_grid2(L, 256);
            switch (set_for_mask->W) {
                case 1: KC_LAUNCH((em_mask_kernel<1>), dim3(gm), dim3(256), 0, ctx->stream, (const char *)out, L, k, complements, kc_view<1>(set_for_mask), mask); break;
                case 2: KC_LAUNCH((em_mask_kernel<2>), dim3(gm), dim3(256), 0, ctx->stream, (const char *)out, L, k, complements, kc_view<2>(set_for_mask), mask); break;
                default: KC_LAUNCH((em_mask_kernel<4>), dim3(gm), dim3(256), 0, ctx->stream, (const char *)out, L, k, complements, kc_view<4>(set_for_mask), mask); break;
            }
            if (hipGetLastError() != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "mask kernel launch failed");
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    kce_rank_release(ctx, &rk);
    if (rc != KC_OK) { kc_release(ctx, out); kc_release(ctx, mask); return rc; }
    *d_out = out;
    if (d_mask) *d_mask = mask;
    *length = L;
    return KC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// maskopt -t max-one / min-one (masks.h:40-78): position p (k-mer start) is ON iff its canonical k-mer is in the set of
// k-mers that start at an upper-case position (AddKMers(case_sensitive=true), masks.h:243); min-one keeps only the
// first occurrence of each k-mer (the reference deletes the k-mer after its first hit, masks.h:60-62).
template <int W, int PASS>
__global__ void __launch_bounds__(256) mo_kernel(const char *__restrict__ ms, u64 len, int k, int complements, int minimize, SetView<W> set,
                                                 u64 *firstpos, char *out, u64 *bad) {
    u64 p = kc_gtid();
    if (p >= len) return;
    char c = ms[p];
    bool upper = c <= 'Z';
    if (p + (u64)k > len) {   // trailing k-1 characters are switched off (masks.h:70-72)
        if (PASS == 1) out[p] = upper ? (char)(c + ('a' - 'A')) : c;
        return;
    }
    Key<W> x = key_zero<W>();
    bool invalid = false;
    for (int i = 0; i < k; ++i) {
        u32 ch = (u32)(u8)ms[p + i];
        u32 up = ch & 0xDFu;
        invalid = invalid || !((up == 'A') | (up == 'C') | (up == 'G') | (up == 'T'));
        u32 code = (up >> 1) & 3u;
        code ^= code >> 1;
        x = key_append(x, (u64)code, k);
    }
    if (invalid) { if (PASS == 0 || !minimize) atomicExch((ull *)bad, 1ULL); if (PASS == 1) out[p] = c; return; }   // (max-one runs pass 1 only)
    if (complements) { Key<W> r = key_rc(x, k); if (key_lt(r, x)) x = r; }
    u64 slot = set_find(set, x);
    bool on = slot != KC_NONE64;
    if (PASS == 0) {
        if (on && minimize) atomicMin((ull *)&firstpos[slot], (ull)p);
        return;
    }
    if (on && minimize) on = (firstpos[slot] == p);
    out[p] = on ? (upper ? c : (char)(c - ('a' - 'A'))) : (upper ? (char)(c + ('a' - 'A')) : c);
}

// count of the bytes equal to `what` (the record separator '\n' of a tokenised FASTA: more than one record)
__global__ void __launch_bounds__(256) mo_count_byte_kernel(const char *__restrict__ s, u64 len, char what, u64 *out) {
    u64 c = 0;
    for (u64 i = kc_gtid(); i < len; i += (u64)gridDim.x * gridDim.y * blockDim.x) c += s[i] == what ? 1 : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd((ull *)out, (ull)c);
}
// d_ms (device, len characters) -> d_out (device, len characters); *bad = an invalid character was met (the output is still complete:
// such positions keep their character, masks.h:75-77 reports after printing)
template <int W> static int maskopt_dev(kc_ctx *ctx, const char *d_ms, u64 len, int k, int complements, int minimize, char *d_out, u64 *hbad) {
    KcScope scope(ctx);
    kc_set *set = nullptr;
    KC_TRY(kc_set_create(ctx, k, complements, len, &set));
    struct SetGuard { kc_set *s; ~SetGuard() { kc_set_destroy(s); } } sg{set};
    u64 *firstpos = nullptr, *bad = nullptr;
    KC_TRY(scope.alloc(&bad, 1));
    KC_TRY(kc_set_add_sequence_device(set, d_ms, len, 1));
    if (minimize) {
        KC_TRY(scope.alloc(&firstpos, set->cap));
        KC_HIP(ctx, hipMemsetAsync(firstpos, 0xff, set->cap * 8, ctx->stream));
    }
    KC_HIP(ctx, hipMemsetAsync(bad, 0, 8, ctx->stream));
    SetView<W> sv = kc_view<W>(set);
    const dim3 g = kc_grid2(len, 256);
    // (pass 0 finds the first occurrence of every k-mer: only min-one needs it)
    if (minimize) KC_LAUNCH((mo_kernel<W, 0>), dim3(g), dim3(256), 0, ctx->stream, d_ms, len, k, complements, minimize, sv, firstpos, d_out, bad);
    KC_LAUNCH((mo_kernel<W, 1>), dim3(g), dim3(256), 0, ctx->stream, d_ms, len, k, complements, minimize, sv, firstpos, d_out, bad);
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipMemcpyAsync(hbad, bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}
static int maskopt_dev_any(kc_ctx *ctx, const char *d_ms, u64 len, int k, int complements, int minimize, char *d_out, u64 *hbad) {
    switch (kc_words_for_k(k)) {
        case 1: return maskopt_dev<1>(ctx, d_ms, len, k, complements, minimize, d_out, hbad);
        case 2: return maskopt_dev<2>(ctx, d_ms, len, k, complements, minimize, d_out, hbad);
        default: return maskopt_dev<4>(ctx, d_ms, len, k, complements, minimize, d_out, hbad);
    }
}

int kce_maskopt(kc_ctx *ctx, const char *ms, u64 len, int k, int complements, int minimize, char *out) {
    KcScope scope(ctx);
    char *d_ms = nullptr, *d_out = nullptr;
    KC_TRY(scope.alloc(&d_ms, len + 64)); KC_TRY(scope.alloc(&d_out, len + 64));
    KC_HIP(ctx, hipMemcpyAsync(d_ms, ms, len, hipMemcpyHostToDevice, ctx->stream));
    u64 hbad = 0;
    KC_TRY(maskopt_dev_any(ctx, d_ms, len, k, complements, minimize, d_out, &hbad));
    KC_HIP(ctx, hipMemcpyAsync(out, d_out, len, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (hbad) return kc_fail(ctx, KC_E_INVALID_CHAR, "Masked superstring contains invalid characters.");
    return KC_OK;
}

// The same with the masked superstring already on the device (kc_read_fasta_device) and the result staying there (kc_result_write_*):
// what `maskopt` on a multi-gigabyte file needs -- the host-string form moves 2 x len bytes through pageable memory.  KC_E_ARG with
// "single FASTA record" when the sequence holds a record separator (masks.h:258: the caller then takes the host parser, which treats the
// first record as the reference does); KC_E_INVALID_CHAR comes WITH a result (masks.h:75-77 prints before it reports).
extern "C" int kc_maskopt_device(kc_ctx *ctx, const void *d_ms, uint64_t len, int k, int complements, int minimize, kc_result **out) {
    if (!ctx || !d_ms || !out) return KC_E_ARG;
    *out = nullptr;
    if (k < 1 || k > 127) return kc_fail(ctx, KC_E_ARG, "k must be in 1..127");
    if (len < (u64)k) return kc_fail(ctx, KC_E_ARG, "masked superstring shorter than k");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    {
        // (the device tokeniser writes one separator per header line, so the sequence of a single record begins with one: skipped)
        KcScope scope(ctx);
        u64 *cnt = nullptr, h = 0;
        char first = 0;
        KC_TRY(scope.alloc(&cnt, 1));
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
        KC_LAUNCH(mo_count_byte_kernel, dim3(1024, 4), dim3(256), 0, ctx->stream, (const char *)d_ms, len, '\n', cnt);
        KC_HIP(ctx, hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipMemcpyAsync(&first, d_ms, 1, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (first == '\n') { d_ms = (const char *)d_ms + 1; --len; --h; }
        if (h) return kc_fail(ctx, KC_E_ARG, "Expecting only a single FASTA record -- the masked superstring.");
        if (len < (u64)k) return kc_fail(ctx, KC_E_ARG, "masked superstring shorter than k");
    }
    kc_result *r = new kc_result();
    r->ctx = ctx; r->length = len; r->slice_off = 0; r->slice_len = len;
    int rc = kc_alloc(ctx, &r->d_out, len + 64);
    u64 hbad = 0;
    if (rc == KC_OK) rc = maskopt_dev_any(ctx, (const char *)d_ms, len, k, complements ? 1 : 0, minimize ? 1 : 0, r->d_out, &hbad);
    if (rc != KC_OK) { kc_result_destroy(r); return rc; }
    *out = r;
    if (hbad) return kc_fail(ctx, KC_E_INVALID_CHAR, "Masked superstring contains invalid characters.");
    return KC_OK;
}

extern "C" int kc_maskopt_ones(kc_ctx *ctx, const char *ms, uint64_t len, int k, int complements, int minimize, char *out) {
    if (!ctx || !ms || !out) return KC_E_ARG;
    if (k < 1 || k > 127) return kc_fail(ctx, KC_E_ARG, "k must be in 1..127");
    if (len < (u64)k) return kc_fail(ctx, KC_E_ARG, "masked superstring shorter than k");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    return kce_maskopt(ctx, ms, len, k, complements ? 1 : 0, minimize ? 1 : 0, out);
}
