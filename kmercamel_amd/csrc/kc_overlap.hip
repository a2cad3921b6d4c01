// Phase 2 of the BIGREEDY path on gfx950: greedy largest-overlap Hamiltonian path over the simplitigs.
// Replaces OverlapHamiltonianPath / OverlapHamiltonianPathSparse (global.h:43-133, global_sparse.h:43-132) with
// byte-identical edge_from / overlaps arrays.
//
// The reference scans every vertex 16 times per overlap length d (MEMORY_REDUCTION_FACTOR, global.h:33,64-66) against
// a hash map of 1/16 of the prefixes.  Here each level d does
//   1. bucket : insert the d-mer prefix keys of all active prefix vertices into a level hash table in HBM (CAS claim),
//               probe it with the d-mer suffix keys of all active suffix vertices, flag keys seen from both sides;
//   2. compact: ordered stream compaction of the vertices whose key was seen from both sides (the "hits"), only for
//               the representative key of each {key, RC(key)} pair;
//   3. sort   : stable radix sort of the hits by key (ids stay ascending inside a key);
//   4. match  : one thread per key bucket replays the reference's event order for that bucket pair
//               (kc_overlap_logic.h), buckets run in parallel;
//   5. verify : path ends of merged paths are recomputed from the level's new edges; a cycle can only appear when two
//               buckets extended the same path in one level and the stale cycle test let an edge through -- the
//               earliest such event is blacklisted, the level is undone and replayed (exactness argument in DESIGN.md);
//   6. shrink : ordered compaction of the active lists.
// Levels without hits (almost all d > log4 V) cost two streaming kernels.
#include <chrono>
#include "kc_internal.h"
#include "kc_comm.h"
#include "kc_overlap_logic.h"
#include "kc_overlap_lds.h"
#include "kc_scan_dev.h"

#define LV_EMPTY (~(u64)0)
#define KC_CNT_BLOCKS 256
#define KC_LDS_MIN 12      /* buckets with fewer members are replayed by a single thread straight from global memory */
#define KC_LDS_MID 128     /* the size class of most listed buckets: 5 KB, 32 workgroups to a CU */
#define KC_LDS_MAX 1200    /* 39 B of LDS per member: 46 KB */
#define KC_LDS_HUGE 3800   /* the few buckets beyond that (d <= 2: one or two buckets hold every vertex that is left) get a CU's whole LDS: 145 KB */

template <int W> struct LevelTable {
    u64 *slot;     // W==1: the d-mer key; W>1: a representative prefix vertex (its key is re-derived from pre[])
    u32 *flags;    // INVERTED (so that one 0xff fill clears slots and flags): bit1 clear = an active suffix vertex has this key
    u64 mask;      // capacity - 1 (power of two)
};

template <int W> __device__ __forceinline__ Key<W> load_key(const u64 *a, u64 i) {
    Key<W> r;
    for (int j = 0; j < W; ++j) r.w[j] = a[i * W + j];
    return r;
}

// insert-or-find for a prefix vertex j with d-mer key `key`
template <int W>
__device__ __forceinline__ u64 level_insert(const LevelTable<W> &t, const Key<W> &key, u64 j, const u64 *pre, int k, int d) {
    u64 s = key_hash(key) & t.mask;
    while (true) {
        u64 cur = __hip_atomic_load(&t.slot[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == LV_EMPTY) {
            u64 want = (W == 1) ? key.w[0] : j;
            u64 prev = atomicCAS((ull *)&t.slot[s], (ull)LV_EMPTY, (ull)want);
            if (prev == LV_EMPTY) return s;
            cur = prev;
        }
        bool same;
        if constexpr (W == 1) same = (cur == key.w[0]);
        else same = key_eq(key_prefix(load_key<W>(pre, cur), k, d), key);
        if (same) return s;
        s = (s + 1) & t.mask;
    }
}
template <int W>
__device__ __forceinline__ u64 level_find(const LevelTable<W> &t, const Key<W> &key, const u64 *pre, int k, int d) {
    u64 s = key_hash(key) & t.mask;
    while (true) {
        u64 cur = t.slot[s];
        if (cur == LV_EMPTY) return KC_NONE64;
        bool same;
        if constexpr (W == 1) same = (cur == key.w[0]);
        else same = key_eq(key_prefix(load_key<W>(pre, cur), k, d), key);
        if (same) return s;
        s = (s + 1) & t.mask;
    }
}

// vertex end k-mers: pre[v] / suf[v] (accessPrefix / accessSuffix, global.h:16-21)
template <int W>
__global__ void __launch_bounds__(256) ov_init_kernel(const u64 *__restrict__ firstK, const u64 *__restrict__ lastK, u64 S, u64 V, int k,
                                                      u64 *pre, u64 *suf, u64 *edge_from, u8 *overlaps, u64 *pred, u64 *first, u64 *last,
                                                      u64 *first_snap, u64 *last_snap, u64 *actS, u64 *actP, u64 *eidx) {
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    Key<W> p, s;
    if (v < S) { p = load_key<W>(firstK, v); s = load_key<W>(lastK, v); }
    else { p = key_rc(load_key<W>(lastK, v - S), k); s = key_rc(load_key<W>(firstK, v - S), k); }
    for (int j = 0; j < W; ++j) { pre[v * W + j] = p.w[j]; suf[v * W + j] = s.w[j]; }
    edge_from[v] = KC_NONE64; overlaps[v] = 0xff; pred[v] = KC_NONE64;
    first_snap[v] = last_snap[v] = v;
    actS[v] = v; actP[v] = v; eidx[v] = KC_NONE64;
}

// Hash-sharded flow: the overlap keys of a level are divided among the G ranks by their hash (a bucket = all suffix / prefix
// vertices with one key, so every bucket is replayed by exactly one rank); the ranks exchange the edges they found (see
// overlap_path_w).  G == 1: everything is ours.
template <int W> __device__ __forceinline__ bool ov_key_is_ours(const Key<W> &key, u32 G, u32 me) {
    return G <= 1 || (u32)(((key_hash(key) >> 32) * (u64)G) >> 32) == me;
}
template <int W>
__global__ void __launch_bounds__(256) ov_insert_prefix_kernel(LevelTable<W> t, const u64 *__restrict__ actP, u64 nP,
                                                               const u64 *__restrict__ pre, const u64 *__restrict__ pred, int k, int d, u64 *pslot, u32 G, u32 me) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nP) return;
    u64 j = actP[i];
    // (the active lists are compacted lazily -- only when a quarter of their entries have died -- so every sweep checks for itself
    // whether a listed vertex still is an active prefix / suffix vertex)
    if (pred[j] != KC_NONE64) { pslot[i] = KC_NONE64; return; }
    Key<W> key = key_prefix(load_key<W>(pre, j), k, d);
    if (!ov_key_is_ours(key, G, me)) { pslot[i] = KC_NONE64; return; }
    pslot[i] = level_insert(t, key, j, pre, k, d);   // (an occupied slot HAS an active prefix: only the suffix side needs a flag)
}

// representative rule: a {key, RC(key)} pair is handled by the smaller of the two keys
template <int W> __device__ __forceinline__ bool is_representative(const Key<W> &key, int d, int complements) {
    if (!complements || d == 0) return true;
    Key<W> r = key_rc(key, d);
    return !key_lt(r, key);
}

// Counts the hits of a level and lists the first hit_cap of them (unordered; a level with few hits then skips the ordered
// compaction).  One global atomic per WORKGROUP: with one per wavefront (what the compiler makes of a plain atomicAdd on a
// uniform address) the ~80 000 same-address atomics of a 5e6-vertex level cost ~1 ms at ~13 ns each.
// Called by every thread of the workgroup (256 threads).
__device__ __forceinline__ void ov_list_hit(u8 hit, u64 v, u64 *hit_count, u64 *hit_list, u64 hit_cap) {
    __shared__ u32 wcount[4];
    __shared__ u64 wbase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(hit != 0);
    if (lane == 0) wcount[wv] = (u32)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        wbase = tot ? atomicAdd((ull *)hit_count, (ull)tot) : 0;
    }
    __syncthreads();
    if (hit) {
        u64 pos = wbase + (u64)__popcll(m & (((unsigned long long)1 << lane) - 1));
        for (int j = 0; j < wv; ++j) pos += wcount[j];
        if (pos < hit_cap) hit_list[pos] = v;
    }
}

// The same for a thread that handled OV_ITEMS vertices (bit j of `hits`: vertex v[j] is a hit): one reservation per workgroup of
// 256 x OV_ITEMS vertices.  (On reads with coverage nearly every workgroup has a hit at every level: with one vertex per thread the
// ~20 000 same-address atomics of a 5e6-vertex sweep were 0.25 ms of each of the two sweeps of a level.)
#define OV_ITEMS 8
__device__ __forceinline__ void ov_list_hits(u32 hits, const u64 (&v)[OV_ITEMS], u64 *hit_count, u64 *hit_list, u64 hit_cap) {
    __shared__ u32 wcount[4];
    __shared__ u64 wbase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 c = (u32)__popc(hits);
    u32 inc = c;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(inc, dd, 64); if (lane >= dd) inc += o; }
    if (lane == 63) wcount[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        wbase = tot ? atomicAdd((ull *)hit_count, (ull)tot) : 0;
    }
    __syncthreads();
    if (c) {
        u64 pos = wbase + (u64)(inc - c);
        for (int j = 0; j < wv; ++j) pos += wcount[j];
#pragma unroll
        for (int j = 0; j < OV_ITEMS; ++j)
            if ((hits >> j) & 1u) { if (pos < hit_cap) hit_list[pos] = v[j]; ++pos; }
    }
}

// (a workgroup takes 256 x OV_ITEMS consecutive positions of the active list, thread t the positions base + j * 256 + t)
template <int W>
__global__ void __launch_bounds__(256) ov_probe_suffix_kernel(LevelTable<W> t, const u64 *__restrict__ actS, u64 nS,
                                                              const u64 *__restrict__ suf, const u64 *__restrict__ pre, const u64 *__restrict__ edge_from, int k, int d,
                                                              int complements, u8 *hitS, u64 *hit_count, u64 *hit_list, u64 hit_cap, u32 G, u32 me) {
    const u64 base = (u64)blockIdx.x * (256 * OV_ITEMS) + threadIdx.x;
    u32 hits = 0;
    u64 v[OV_ITEMS];
#pragma unroll
    for (int j = 0; j < OV_ITEMS; ++j) {
        const u64 i = base + (u64)j * 256;
        v[j] = 0;
        if (i < nS) {
            v[j] = actS[i];
            Key<W> key = key_suffix(load_key<W>(suf, v[j]), d);
            u64 s = (edge_from[v[j]] == KC_NONE64 && ov_key_is_ours(key, G, me)) ? level_find(t, key, pre, k, d) : KC_NONE64;   // (a listed vertex may have died: lazy compaction)
            u8 hit = 0;
            if (s != KC_NONE64) {
                atomicAnd(&t.flags[s], ~2u);
                if (is_representative(key, d, complements)) hit = 1;
            }
            hitS[i] = hit;
            hits |= (u32)hit << j;
        }
    }
    ov_list_hits(hits, v, hit_count, hit_list, hit_cap);
}

template <int W>
__global__ void __launch_bounds__(256) ov_flag_prefix_kernel(LevelTable<W> t, const u64 *__restrict__ actP, u64 nP,
                                                             const u64 *__restrict__ pre, const u64 *__restrict__ pslot, int k, int d,
                                                             int complements, u8 *hitP, u64 *hit_count, u64 *hit_list, u64 hit_cap) {
    const u64 base = (u64)blockIdx.x * (256 * OV_ITEMS) + threadIdx.x;
    u32 hits = 0;
    u64 v[OV_ITEMS];
#pragma unroll
    for (int j = 0; j < OV_ITEMS; ++j) {
        const u64 i = base + (u64)j * 256;
        v[j] = 0;
        if (i < nP) {
            u8 hit = 0;
            if (pslot[i] != KC_NONE64 && !(t.flags[pslot[i]] & 2u)) {   // (KC_NONE64: the key belongs to another rank)
                Key<W> key = key_prefix(load_key<W>(pre, actP[i]), k, d);
                if (is_representative(key, d, complements)) hit = 1;
            }
            hitP[i] = hit;
            if (hit) v[j] = actP[i];
            hits |= (u32)hit << j;
        }
    }
    ov_list_hits(hits, v, hit_count, hit_list, hit_cap);
}

// ---- levels d >= 16 through ONE table (large inputs).  A hit at level d >= 16 needs the first 16 bases of the prefix to
// equal the 16 bases of the suffix vertex's k-mer that start k-d from its end, so one table of the prefix ends' first
// 16-mers (built once; equal 16-mers chained through `next`) serves every such level: a level is then one sweep over the
// active suffixes (window lookup, exact d-mer compare with the few chained candidates) instead of table fill + 5e6 CAS
// inserts + probe sweep + flag sweep.  Most of these levels have no hit at all.  Same hit sets as the per-level table.
struct T16 { u64 *key; u32 *head; u32 *next; u64 mask; };   // key: the 16-mer (32 bits), ~0 = empty; head / next: 0xffffffff = none
#define KC_T16_CHAIN_MAX 16   /* longer chains (a 16-mer shared by many simplitig ends): this and the later levels build tables of their own */
template <int W>
__global__ void __launch_bounds__(256) ov_t16_build_kernel(T16 t, const u64 *__restrict__ pre, u64 V, int k, u64 *distinct) {
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u8 fresh = 0;   // this vertex brought a 16-mer the table did not have yet
    if (v < V) {
        const u64 k16 = key_prefix(load_key<W>(pre, v), k, 16).w[0] & 0xffffffffull;
        u64 s = kc_mix64(k16) & t.mask;
        while (true) {
            u64 cur = __hip_atomic_load(&t.key[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == ~(u64)0) { const u64 prev = atomicCAS((ull *)&t.key[s], ~0ULL, (ull)k16); if (prev == ~(u64)0) { fresh = 1; cur = k16; } else cur = prev; }
            if (cur == k16) break;
            s = (s + 1) & t.mask;
        }
        t.next[v] = atomicExch(&t.head[s], (u32)v);
    }
    ov_list_hit(fresh, 0, distinct, nullptr, 0);   // number of distinct 16-mers: the host decides from it whether the table pays
}
template <int W>
__global__ void __launch_bounds__(256) ov_probe16_kernel(T16 t, const u64 *__restrict__ actS, u64 nS, const u64 *__restrict__ suf, const u64 *__restrict__ pre,
                                                         const u64 *__restrict__ pred, const u64 *__restrict__ edge_from, int k, int d, int complements, u8 *hitS, u32 *markP, u64 *cnt, u64 *hs_list,
                                                         u64 *hp_list, u64 hit_cap, u32 G, u32 me) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u8 hit = 0;
    u64 u = 0;
    if (i < nS) {
        u = actS[i];
        const Key<W> key = key_suffix(load_key<W>(suf, u), d);
        if (edge_from[u] == KC_NONE64 && ov_key_is_ours(key, G, me)) {   // (a listed vertex may have died: lazy compaction)
            const u64 k16 = key_prefix(key, d, 16).w[0] & 0xffffffffull;
            u64 s = kc_mix64(k16) & t.mask;
            u64 cur;
            while ((cur = t.key[s]) != ~(u64)0 && cur != k16) s = (s + 1) & t.mask;
            if (cur == k16) {
                const bool rep = is_representative(key, d, complements);
                bool found = false;
                int steps = 0;
                for (u32 v = t.head[s]; v != 0xffffffffu; v = t.next[v]) {
                    if (++steps > KC_T16_CHAIN_MAX) { if (__hip_atomic_load(cnt + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) cnt[4] = 1; break; }
                    if (pred[v] != KC_NONE64) continue;   // no longer an active prefix vertex
                    if (!key_eq(key_prefix(load_key<W>(pre, v), k, d), key)) continue;
                    found = true;
                    if (rep && atomicExch(&markP[v], 1u) == 0u) {   // first suffix to reach this prefix vertex
                        // few hits (the usual case at these levels): list and count them here.  Many (overlapping reads): one add per
                        // hit on one word would serialise at ~12 ns each, so past the list's capacity only a flag is raised and the
                        // host counts the marks in a sweep of their own (ov_marks_to_hits_kernel).
                        if (__hip_atomic_load(cnt + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < hit_cap) {
                            const u64 pos = atomicAdd((ull *)(cnt + 3), 1ULL);
                            if (pos < hit_cap) hp_list[pos] = v;
                        } else if (__hip_atomic_load(cnt + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) cnt[5] = 1;   // (millions of plain stores to one word serialise too)
                    }
                }
                hit = (found && rep) ? 1 : 0;
            }
        }
        hitS[i] = hit;
    }
    ov_list_hit(hit, u, cnt + 1, hs_list, hit_cap);
}

// Counts, for the levels d0, d0 - 1, ... d0 - nlev + 1 (all >= 16) at once, the active suffix vertices that would hit at that level
// if nothing changed in between: hist[j] for level d0 - j.  Edges only ever REMOVE vertices from the active sets, so a level that
// counts zero here has no hit when its turn comes, whatever the levels before it did: the level loop skips it without a launch or
// a read-back (k = 127 on 10^5 reads: 110 such levels in a row).  Levels that count something are probed as usual.
template <int W>
__global__ void __launch_bounds__(256) ov_count16_kernel(T16 t, const u64 *__restrict__ actS, u64 nS, const u64 *__restrict__ suf, const u64 *__restrict__ pre,
                                                         const u64 *__restrict__ pred, const u64 *__restrict__ edge_from, int k, int d0, int nlev, u64 *hist,
                                                         u32 G, u32 me) {
    __shared__ u32 lh[128];
    if (threadIdx.x < 128) lh[threadIdx.x] = 0;
    __syncthreads();
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nS) {
        const u64 u = actS[i];
        if (edge_from[u] == KC_NONE64) {
            const Key<W> sfx = load_key<W>(suf, u);
            for (int j = 0; j < nlev; ++j) {
                const int d = d0 - j;
                const Key<W> key = key_suffix(sfx, d);
                if (!ov_key_is_ours(key, G, me)) continue;
                const u64 k16 = key_prefix(key, d, 16).w[0] & 0xffffffffull;
                u64 s = kc_mix64(k16) & t.mask;
                u64 cur;
                while ((cur = t.key[s]) != ~(u64)0 && cur != k16) s = (s + 1) & t.mask;
                if (cur != k16) continue;
                bool found = false;
                int steps = 0;
                for (u32 v = t.head[s]; v != 0xffffffffu && !found; v = t.next[v]) {
                    if (++steps > KC_T16_CHAIN_MAX) { found = true; break; }   // (a long chain: let the level's own probe decide)
                    if (pred[v] != KC_NONE64) continue;
                    found = key_eq(key_prefix(load_key<W>(pre, v), k, d), key);
                }
                if (found) atomicAdd(&lh[j], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)nlev && lh[threadIdx.x]) atomicAdd((ull *)&hist[threadIdx.x], (ull)lh[threadIdx.x]);
}

// hitP (by position in actP) and the exact number of prefix hits from the per-vertex marks of ov_probe16_kernel: only needed
// when the prefix hits outgrow their list
__global__ void __launch_bounds__(256) ov_marks_to_hits_kernel(const u64 *__restrict__ actP, u64 nP, const u32 *__restrict__ markP, u8 *hitP, u64 *count,
                                                               u64 *hp_list, u64 hit_cap) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u8 hit = 0;
    u64 v = 0;
    if (i < nP) { v = actP[i]; hit = markP[v] ? 1 : 0; hitP[i] = hit; }
    ov_list_hit(hit, v, count, hp_list, hit_cap);
}

template <int W, bool SUFFIX>
__global__ void __launch_bounds__(256) ov_hit_keys_kernel(const u64 *__restrict__ ids, u64 n, const u64 *__restrict__ ends, int k, int d,
                                                          u64 *keys) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Key<W> e = load_key<W>(ends, ids[i]);
    Key<W> key = SUFFIX ? key_suffix(e, d) : key_prefix(e, k, d);
    for (int j = 0; j < W; ++j) keys[i * W + j] = key.w[j];
}

__global__ void __launch_bounds__(256) ov_iota_kernel(u64 *a, u64 n) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = t;
}
__global__ void __launch_bounds__(256) ov_gather_word_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ perm, u64 n, int W,
                                                             int word, u64 *out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = keys[perm[t] * W + word];
}
__global__ void __launch_bounds__(256) ov_gather_rows_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ ids,
                                                             const u64 *__restrict__ perm, u64 n, int W, u64 *keys_out, u64 *ids_out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 p = perm[t];
    for (int j = 0; j < W; ++j) keys_out[t * W + j] = keys[p * W + j];
    ids_out[t] = ids[p];
}

template <int W> __device__ __forceinline__ int cmp_rows(const u64 *a, u64 i, const Key<W> &key) {  // -1 a<key, 0 eq, 1 a>key
    Key<W> x = load_key<W>(a, i);
    if (key_lt(x, key)) return -1;
    return key_eq(x, key) ? 0 : 1;
}

// first index >= start whose row is greater than key, in sorted rows none of which from `start` on is smaller than key
template <int W> __device__ __forceinline__ u64 ov_run_end(const u64 *a, u64 start, u64 n, const Key<W> &key) {
    u64 lo = start, step = 1;
    while (lo + step <= n && cmp_rows<W>(a, lo + step - 1, key) <= 0) { lo += step; step <<= 1; }
    u64 hi = lo + step - 1 < n ? lo + step - 1 : n;   // (row hi is greater than key, or hi = n)
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (cmp_rows<W>(a, mid, key) <= 0) lo = mid + 1; else hi = mid; }
    return lo;
}

// one thread per sorted suffix hit; the first hit of a key run replays the whole bucket (pair)
template <int W>
__global__ void __launch_bounds__(64) ov_match_kernel(OvArgs g, const u64 *__restrict__ hsk, const u64 *__restrict__ hsv, u64 nHS,
                                                      const u64 *__restrict__ hpk, const u64 *__restrict__ hpv, u64 nHP, u64 *scrS, u64 *scrP,
                                                      u64 *big, u64 big_cap, u64 *big_count, u64 *huge, u64 huge_cap, u64 *huge_count) {
    __shared__ __attribute__((aligned(16))) unsigned char lane_lds[64 * KC_LANE_STRIDE];
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nHS) return;
    if (g.abort2 && (g.abort2[0] | g.abort2[1])) return;   // a bucketed sort overflowed: the host redoes the level's sorts
    Key<W> key = load_key<W>(hsk, t);
    if (t > 0 && key_eq(load_key<W>(hsk, t - 1), key)) return;
    // bucket ends: the run of equal keys is followed in doubling steps and then bisected (a linear scan is one dependent global load per
    // member: 0.3 ms for the single bucket of d = 0; a bisection of the whole list is 20 dependent loads for a run of two)
    const u64 a1 = ov_run_end<W>(hsk, t + 1, nHS, key);
    u64 lo = 0, hi = nHP;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (cmp_rows<W>(hpk, mid, key) < 0) lo = mid + 1; else hi = mid; }
    const u64 b0 = lo;
    const u64 b1 = ov_run_end<W>(hpk, b0, nHP, key);
    if (b1 == b0) return;
#ifdef KC_EXPERIMENTS
    if (g.exp & 1) return;
#endif
    int mode = KC_MODE_UNI;
    if (g.complements) mode = (g.d == 0 || key_eq(key_rc(key, g.d), key)) ? KC_MODE_PAL : KC_MODE_PAIR;
    const u64 members = (a1 - t) + (b1 - b0);
    if (members >= (u64)g.lds_min && members <= KC_LDS_MAX) {   // medium buckets: replayed out of LDS by a wavefront of their own
        u64 pos = atomicAdd((ull *)big_count, 1ULL);
        if (pos < big_cap) {
            u64 *dsc = big + pos * 5;
            dsc[0] = t; dsc[1] = a1; dsc[2] = b0; dsc[3] = b1; dsc[4] = (u64)mode;
            return;
        }
    }
    if (members > KC_LDS_MAX && members <= KC_LDS_HUGE) {
        u64 pos = atomicAdd((ull *)huge_count, 1ULL);
        if (pos < huge_cap) {
            u64 *dsc = huge + pos * 5;
            dsc[0] = t; dsc[1] = a1; dsc[2] = b0; dsc[3] = b1; dsc[4] = (u64)mode;
            return;
        }
    }
    if (members <= KC_LANE_MAX && !(g.lane_off)) {   // (nearly all of them: two to four members)
        kc_match_bucket_lane(g, hsv + t, (u32)(a1 - t), hpv + b0, (u32)(b1 - b0), mode, lane_lds + (threadIdx.x & 63) * KC_LANE_STRIDE, false, g.tag_base + (u32)t);
        return;
    }
    kc_match_bucket(g, hsv + t, a1 - t, hpv + b0, b1 - b0, mode, g.tag_base + (u32)t, scrS + t, scrP + b0);
}

// one wavefront per medium bucket (descriptors listed by ov_match_kernel); the grid is an upper bound, surplus blocks exit
// (lo, hi]: the bucket sizes this launch takes -- the LDS a workgroup asks for is its launch's largest bucket, and most listed buckets
// have a few dozen members: those run 32 to a CU instead of 3)
__global__ void __launch_bounds__(64) ov_match_lds_kernel(OvArgs g, const u64 *__restrict__ hsv, const u64 *__restrict__ hpv,
                                                          const u64 *__restrict__ big, u64 big_cap, const u64 *__restrict__ big_count, u32 lo, u32 hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char kc_lds[];
    u64 nb = *big_count;
    if (nb > big_cap) nb = big_cap;
    if ((u64)blockIdx.x >= nb) return;   // (no descriptors when the match kernel stood down: abort2)
    const u64 *dsc = big + (u64)blockIdx.x * 5;
    const u64 members = (dsc[1] - dsc[0]) + (dsc[3] - dsc[2]);
    if (members <= lo || members > hi) return;
    kc_match_bucket_lds(g, hsv + dsc[0], (u32)(dsc[1] - dsc[0]), hpv + dsc[2], (u32)(dsc[3] - dsc[2]), (int)dsc[4], kc_lds);
}

// remember, per vertex, which log entry created its in-edge (used only by the cycle repair)
__global__ void __launch_bounds__(256) ov_index_edges_kernel(const u64 *__restrict__ ne, u64 n, u64 *eidx) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) eidx[ne[t * 5 + 1]] = t;
}

// Path ends after the level: walk back from every merged path's tail over the level's new edges.
template <bool WRITE>
__global__ void __launch_bounds__(256) ov_fix_kernel(const u64 *__restrict__ ne, const u64 *__restrict__ n_ptr, const u64 *__restrict__ edge_from,
                                                     const u64 *__restrict__ pred, u64 *first_snap, u64 *last_snap, u8 *cov, u64 *covered) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n = *n_ptr;   // number of edges the match kernel logged (device-resident: no host round trip in between)
    if (t >= n) return;
    u64 y = ne[t * 5 + 1];
    u64 l0 = last_snap[y];
    if (edge_from[l0] != KC_NONE64) return;   // not the tail of its merged path
    u64 f = y, steps = 0;
    while (steps <= n) {
        u64 xp = pred[f];
        if (xp == KC_NONE64) break;
        if (!WRITE) cov[f] = 1;
        ++steps;
        f = first_snap[xp];
    }
    if (WRITE) { first_snap[l0] = f; last_snap[f] = l0; }
    else atomicAdd((ull *)covered, (ull)steps);
}
// the level is acyclic: write the new path ends and clear the coverage marks in one launch (one thread per logged edge; the two
// jobs touch different arrays)
__global__ void __launch_bounds__(256) ov_fix_write_clear_kernel(const u64 *__restrict__ ne, const u64 *__restrict__ n_ptr, const u64 *__restrict__ edge_from,
                                                                 const u64 *__restrict__ pred, u64 *first_snap, u64 *last_snap, u8 *cov) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 n = *n_ptr;
    if (t >= n) return;
    const u64 y = ne[t * 5 + 1];
    cov[y] = 0;
    const u64 l0 = last_snap[y];
    if (edge_from[l0] != KC_NONE64) return;   // not the tail of its merged path
    u64 f = y, steps = 0;
    while (steps <= n) {
        const u64 xp = pred[f];
        if (xp == KC_NONE64) break;
        ++steps;
        f = first_snap[xp];
    }
    first_snap[l0] = f; last_snap[f] = l0;
}

// cycle repair (rare, one thread): among the cycles left uncovered, find the one whose latest event is earliest, and
// report that event (u, c).  res = {found, part, u, c}
// Pass 1, one thread per new edge: an uncovered edge lies on a cycle; the thread whose target vertex is the smallest of its
// cycle reports the cycle's latest event (part, u) to an atomic minimum.  res[1] = (part << 32) | u, initialised to ~0.
__global__ void __launch_bounds__(256) ov_find_cycle_kernel(const u64 *__restrict__ ne, u64 n, const u64 *__restrict__ first_snap,
                                                            const u64 *__restrict__ eidx, const u8 *__restrict__ cov, u64 *res) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u64 y0 = ne[e * 5 + 1];
    if (cov[y0]) return;
    u64 cur = y0, mp = 0, mu = 0;
    bool any = false;
    do {
        if (cur < y0) return;                         // another thread owns this cycle
        const u64 *ed = ne + eidx[cur] * 5;
        const u64 part = ed[4], eu = ed[2];
        if (!any || part > mp || (part == mp && eu > mu)) { mp = part; mu = eu; any = true; }
        cur = first_snap[ed[0]];
    } while (cur != y0);
    atomicMin((ull *)(res + 1), (ull)((mp << 32) | mu));
}
// Pass 2: the edge logged by that event supplies c.  res = {found, part, u, c}
__global__ void __launch_bounds__(256) ov_pick_event_kernel(const u64 *__restrict__ ne, u64 n, u64 *res) {
    u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const u64 best = res[1];
    if (best == ~(u64)0) return;
    const u64 *ed = ne + e * 5;
    if (ed[4] == (best >> 32) && ed[2] == (best & 0xffffffffull) && ed[0] == ed[2]) {   // the event's own edge (not its mirror)
        res[0] = 1; res[2] = ed[2]; res[3] = ed[3];
    }
}

__global__ void __launch_bounds__(256) ov_undo_kernel(const u64 *__restrict__ ne, u64 n, u64 *edge_from, u8 *overlaps, u64 *pred, u64 *eidx,
                                                      u8 *cov) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 x = ne[t * 5], y = ne[t * 5 + 1];
    edge_from[x] = KC_NONE64; overlaps[x] = 0xff; pred[y] = KC_NONE64; eidx[y] = KC_NONE64; cov[y] = 0;
}
__global__ void __launch_bounds__(256) ov_clear_cov_kernel(const u64 *__restrict__ ne, u64 n, u8 *cov) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) cov[ne[t * 5 + 1]] = 0;
}


// the 64 shards of the new-edge log -> the dense log `ne`; cnt[0] = number of entries (what the verification kernels read)
__global__ void __launch_bounds__(256) ov_gather_log_kernel(const u64 *__restrict__ ne_sh, const u64 *__restrict__ shcur, u64 shcap, u64 *ne, u64 *cnt) {
    __shared__ u64 pre[65];
    if (threadIdx.x == 0) {
        u64 a = 0;
        for (int s = 0; s < 64; ++s) { u64 c = shcur[s * 16]; if (c > shcap) c = shcap; pre[s] = a; a += c; }
        pre[64] = a;
    }
    __syncthreads();
    const int sh = blockIdx.y;
    const u64 n = (pre[sh + 1] - pre[sh]) * 5;
    const u64 *src = ne_sh + (u64)sh * shcap * 5;
    u64 *dst = ne + pre[sh] * 5;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) cnt[0] = pre[64];
}

// hash-sharded flow: the edges the other ranks found at this level (log entries outside [own_lo, own_hi))
__global__ void __launch_bounds__(256) ov_apply_remote_kernel(const u64 *__restrict__ ne, u64 n, u64 own_lo, u64 own_hi, int d, u64 *edge_from, u8 *overlaps,
                                                              u64 *pred) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || (t >= own_lo && t < own_hi)) return;
    const u64 x = ne[t * 5], y = ne[t * 5 + 1];
    edge_from[x] = y; overlaps[x] = (u8)d; pred[y] = x;
}

template <bool SUFFIX>
__global__ void __launch_bounds__(256) ov_still_active_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ edge_from,
                                                              const u64 *__restrict__ pred, u8 *keep) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 v = act[t];
    keep[t] = SUFFIX ? (edge_from[v] == KC_NONE64) : (pred[v] == KC_NONE64);
}

// ---------------------------------------------------------------------------------------------------------
// GROUPED LEVELS: the level loop without the host in it.  A level of the host-driven loop needs the hit
// counts on the host twice (to size its compactions, sorts and match launches) and ~27 launches; here the members of a key
// bucket are brought together by the level table itself -- every slot counts its prefix and suffix vertices, the slot's
// creator reserves the bucket's stretch of the member arrays, the vertices scatter themselves into it through the slot's
// cursors, and the thread (or wavefront) that replays the bucket puts its few members in order first -- so no compaction, no
// key extraction and no sort is launched, every kernel takes its sizes from a block of device words (OVG_*), and the host
// enqueues ALL remaining levels back to back and looks at the outcome once.  A level that needs what this path does not do --
// a repair (a cycle slipped through), a bucket beyond a wavefront's LDS, a count beyond 16 bits -- raises OVG_STATUS: every
// later kernel returns at once, and the host runs that level through the host-driven loop and resumes here.
// The replay is the same code (kc_match_bucket / kc_match_bucket_lds) on the same ascending member lists: identical results.
enum { OVG_LENS = 0, OVG_LENP, OVG_LIVES, OVG_LIVEP, OVG_NB, OVG_TOTS, OVG_TOTP, OVG_STATUS, OVG_MASK, OVG_COVERED, OVG_NE, OVG_BIG, OVG_HUGE, OVG_CUR, OVG_COMPACT,
       OVG_UNSORTED, OVG_NEWS, OVG_NEWP, OVG_LEVELS_DONE, OVG_EDGES, OVG_WORDS = 32 };
#define OVG_GRID 1024
#define OVG_NONE32 0xffffffffu
#define OVG_REASON_REPAIR 1
#define OVG_REASON_BIG 2      /* a bucket with more members than a wavefront replays out of LDS, or a slot count beyond 16 bits */
#define OVG_REASON_LOG 3
struct OvgBuf {
    u64 *act[4];       // active suffix lists (0, 1) and prefix lists (2, 3): st[OVG_CUR] says which of each pair is current
    u64 *pslot;        // per prefix list position: level-table slot | creator << 63, or ~0
    u32 *sslot;        // per suffix list position: slot or OVG_NONE32
    u32 *cnt2;         // per slot: suffix count << 16 | prefix count; after the buckets are laid out: the fill cursors
    u32 *boS, *boP;    // per slot: first member of its bucket in memS / memP, or OVG_NONE32
    u64 *memS, *memP, *scrS, *scrP;   // members / replay scratch
    u64 *descA, *descB;               // per bucket: offS | cS << 32, offP | cP << 32
    u64 *big, *huge;   // descriptors of the buckets replayed by a wavefront
    u64 big_cap, huge_cap, tcap;
};
__device__ __forceinline__ void ovg_poison(u64 *st, int d, int reason) {
    atomicCAS((ull *)&st[OVG_STATUS], 0ULL, (ull)((u64)(d + 1) | ((u64)reason << 8)));
}
// The level's book-keeping rides on the kernels that are launched anyway (round 6: four single-thread launches per level -- begin, decide,
// end, swap -- were a fifth of the launches of an input of 10^8 k-mers, whose grouped levels are bound by launch latency):
//  * ovg_clear_kernel: thread 0 of block 0 first finishes the PREVIOUS level (the swap of the active lists after a compaction) and sets
//    this level's sizes; every block derives the table size itself from the live count, which the swap does not touch;
//  * ovg_write_kernel: every thread takes the verdict of the level (acyclic iff every new edge was covered) from the same words,
//    thread 0 of block 0 records it and closes the level's accounts.
__device__ __forceinline__ u64 ovg_level_cap(const u64 *st, u64 tcap) {
    u64 lcap = 64;
    while (lcap < 2 * st[OVG_LIVEP] + 16 && lcap < tcap) lcap <<= 1;
    return lcap;
}
__device__ __forceinline__ void ovg_swap_lists(u64 *st) {   // one thread: a compaction of the level before has filled the other pair of lists
    if (!st[OVG_COMPACT]) return;
    st[OVG_LENS] = st[OVG_NEWS]; st[OVG_LENP] = st[OVG_NEWP];
    st[OVG_CUR] ^= 1; st[OVG_UNSORTED] = 1; st[OVG_COMPACT] = 0;
}
template <int W> __global__ void __launch_bounds__(256) ovg_clear_kernel(u64 *st, LevelTable<W> t, u32 *cnt2, u64 *shcur, u64 tcap) {
    if (st[OVG_STATUS]) return;
    const u64 n = ovg_level_cap(st, tcap);
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            ovg_swap_lists(st);
            st[OVG_NB] = 0; st[OVG_TOTS] = 0; st[OVG_TOTP] = 0; st[OVG_COVERED] = 0; st[OVG_NE] = 0; st[OVG_BIG] = 0; st[OVG_HUGE] = 0;
            st[OVG_MASK] = n - 1;
        }
        for (int i = threadIdx.x; i < 65 * 16; i += blockDim.x) shcur[i] = 0;
    }
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) { t.slot[i] = LV_EMPTY; cnt2[i] = 0; }
}
// (behind the last level of a run: the swap that the next level's clear would have done)
__global__ void ovg_finish_kernel(u64 *st) {
    if (st[OVG_STATUS]) return;
    ovg_swap_lists(st);
}
// insert-or-find that also says whether this call created the slot
template <int W>
__device__ __forceinline__ u64 level_insert_c(const LevelTable<W> &t, const Key<W> &key, u64 j, const u64 *pre, int k, int d, bool *created) {
    u64 s = key_hash(key) & t.mask;
    *created = false;
    while (true) {
        u64 cur = __hip_atomic_load(&t.slot[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == LV_EMPTY) {
            u64 want = (W == 1) ? key.w[0] : j;
            u64 prev = atomicCAS((ull *)&t.slot[s], (ull)LV_EMPTY, (ull)want);
            if (prev == LV_EMPTY) { *created = true; return s; }
            cur = prev;
        }
        bool same;
        if constexpr (W == 1) same = (cur == key.w[0]);
        else same = key_eq(key_prefix(load_key<W>(pre, cur), k, d), key);
        if (same) return s;
        s = (s + 1) & t.mask;
    }
}
template <int W>
__global__ void __launch_bounds__(256) ovg_insert_kernel(u64 *st, LevelTable<W> t, OvgBuf b, const u64 *__restrict__ pre, const u64 *__restrict__ pred, int k, int d) {
    if (st[OVG_STATUS]) return;
    t.mask = st[OVG_MASK];
    const u64 n = st[OVG_LENP];
    const u64 *actP = b.act[2 + (st[OVG_CUR] & 1)];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 j = actP[i];
        if (pred[j] != KC_NONE64) { b.pslot[i] = KC_NONE64; continue; }   // (no longer an active prefix vertex: the lists are compacted lazily)
        const Key<W> key = key_prefix(load_key<W>(pre, j), k, d);
        bool created;
        const u64 s = level_insert_c(t, key, j, pre, k, d, &created);
        const u32 old = atomicAdd(&b.cnt2[s], 1u);
        if ((old & 0xffffu) >= 0xfff0u) ovg_poison(st, d, OVG_REASON_BIG);
        b.pslot[i] = s | ((u64)created << 63);
    }
}
template <int W>
__global__ void __launch_bounds__(256) ovg_probe_kernel(u64 *st, LevelTable<W> t, OvgBuf b, const u64 *__restrict__ suf, const u64 *__restrict__ pre,
                                                        const u64 *__restrict__ edge_from, int k, int d) {
    if (st[OVG_STATUS]) return;
    t.mask = st[OVG_MASK];
    const u64 n = st[OVG_LENS];
    const u64 *actS = b.act[st[OVG_CUR] & 1];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = actS[i];
        u32 sl = OVG_NONE32;
        if (edge_from[v] == KC_NONE64) {
            const Key<W> key = key_suffix(load_key<W>(suf, v), d);
            const u64 s = level_find(t, key, pre, k, d);
            if (s != KC_NONE64) {
                const u32 old = atomicAdd(&b.cnt2[s], 1u << 16);
                if ((old >> 16) >= 0xfff0u) ovg_poison(st, d, OVG_REASON_BIG);
                sl = (u32)s;
            }
        }
        b.sslot[i] = sl;
    }
}
// the creator of a slot lays out its bucket: the key's representative (of {key, RC(key)}) with members on both sides
template <int W>
__global__ void __launch_bounds__(256) ovg_buckets_kernel(u64 *st, LevelTable<W> t, OvgBuf b, const u64 *__restrict__ pre, int k, int d, int complements) {
    if (st[OVG_STATUS]) return;
    __shared__ u64 base3[3];
    const u64 n = st[OVG_LENP];
    for (u64 i0 = (u64)blockIdx.x * blockDim.x; i0 < n; i0 += (u64)gridDim.x * blockDim.x) {   // (block-uniform trip count: barriers inside)
        const u64 i = i0 + threadIdx.x;
        u32 cS = 0, cP = 0;
        u64 s = 0;
        bool mine = false, bucket = false;
        if (i < n) {
            const u64 ps = b.pslot[i];
            if (ps != KC_NONE64 && (ps >> 63)) {
                mine = true;
                s = ps & ~((u64)1 << 63);
                const u32 c = b.cnt2[s];
                cS = c >> 16; cP = c & 0xffffu;
                if (cS && cP) {
                    Key<W> key;
                    if constexpr (W == 1) key.w[0] = t.slot[s];
                    else key = key_prefix(load_key<W>(pre, t.slot[s]), k, d);
                    bucket = is_representative(key, d, complements);
                    if (bucket && (u64)cS + cP > KC_LDS_HUGE) ovg_poison(st, d, OVG_REASON_BIG);
                }
            }
        }
        u64 t0, t1, t2;
        const u64 e0 = kc_block_excl_scan(bucket ? 1 : 0, &t0);
        const u64 e1 = kc_block_excl_scan(bucket ? cS : 0, &t1);
        const u64 e2 = kc_block_excl_scan(bucket ? cP : 0, &t2);
        if (threadIdx.x == 0) {
            base3[0] = t0 ? atomicAdd((ull *)&st[OVG_NB], (ull)t0) : 0;
            base3[1] = t1 ? atomicAdd((ull *)&st[OVG_TOTS], (ull)t1) : 0;
            base3[2] = t2 ? atomicAdd((ull *)&st[OVG_TOTP], (ull)t2) : 0;
        }
        __syncthreads();
        if (mine) {
            if (bucket) {
                const u64 bi = base3[0] + e0, oS = base3[1] + e1, oP = base3[2] + e2;
                b.boS[s] = (u32)oS; b.boP[s] = (u32)oP;
                b.descA[bi] = oS | ((u64)cS << 32); b.descB[bi] = oP | ((u64)cP << 32);
            } else b.boS[s] = OVG_NONE32;
            b.cnt2[s] = 0;   // from here on: the fill cursors of the bucket
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) ovg_scatter_kernel(const u64 *__restrict__ st, OvgBuf b) {
    if (st[OVG_STATUS]) return;
    const u64 nS = st[OVG_LENS], nP = st[OVG_LENP];
    const u64 *actS = b.act[st[OVG_CUR] & 1], *actP = b.act[2 + (st[OVG_CUR] & 1)];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nS + nP; i += (u64)gridDim.x * blockDim.x) {
        if (i < nS) {
            const u32 s = b.sslot[i];
            if (s == OVG_NONE32) continue;
            const u32 o = b.boS[s];
            if (o == OVG_NONE32) continue;
            b.memS[(u64)o + (atomicAdd(&b.cnt2[s], 1u << 16) >> 16)] = actS[i];
        } else {
            const u64 ps = b.pslot[i - nS];
            if (ps == KC_NONE64) continue;
            const u64 s = ps & ~((u64)1 << 63);
            if (b.boS[s] == OVG_NONE32) continue;
            b.memP[(u64)b.boP[s] + (atomicAdd(&b.cnt2[s], 1u) & 0xffffu)] = actP[i - nS];
        }
    }
}
__device__ __forceinline__ void ovg_insertion_sort(u64 *a, u32 n) {
    for (u32 i = 1; i < n; ++i) {
        const u64 v = a[i];
        u32 j = i;
        while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; --j; }
        a[j] = v;
    }
}
// one thread per bucket: small ones are put in order and replayed here, the others are handed to a wavefront each
template <int W>
__global__ void __launch_bounds__(64) ovg_match_kernel(OvArgs g, u64 *st, OvgBuf b, const u64 *__restrict__ suf) {
    __shared__ __attribute__((aligned(16))) unsigned char lane_lds[64 * KC_LANE_STRIDE];
    if (st[OVG_STATUS]) return;
    const u64 nb = st[OVG_NB];
    const u32 lds_min = (st[OVG_TOTS] + st[OVG_TOTP] <= 32768) ? 4u : g.lds_min;   // (few hits: even small buckets get a wavefront and LDS)
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < nb; t += (u64)gridDim.x * blockDim.x) {
        const u64 dA = b.descA[t], dB = b.descB[t];
        const u64 oS = dA & 0xffffffffull, oP = dB & 0xffffffffull;
        const u32 cS = (u32)(dA >> 32), cP = (u32)(dB >> 32);
        const Key<W> key = key_suffix(load_key<W>(suf, b.memS[oS]), g.d);
        int mode = KC_MODE_UNI;
        if (g.complements) mode = (g.d == 0 || key_eq(key_rc(key, g.d), key)) ? KC_MODE_PAL : KC_MODE_PAIR;
        const u32 members = cS + cP;
        if (members >= lds_min) {
            const bool hg = members > KC_LDS_MAX;
            const u64 pos = atomicAdd((ull *)&st[hg ? OVG_HUGE : OVG_BIG], 1ULL);
            if (pos < (hg ? b.huge_cap : b.big_cap)) {
                u64 *dsc = (hg ? b.huge : b.big) + pos * 5;
                dsc[0] = oS; dsc[1] = oS + cS; dsc[2] = oP; dsc[3] = oP + cP; dsc[4] = (u64)mode;
                continue;
            }
            ovg_poison(st, g.d, OVG_REASON_BIG);   // (cannot happen with the caller's sizing)
            continue;
        }
        if (members <= KC_LANE_MAX && !(g.lane_off)) {
            kc_match_bucket_lane(g, b.memS + oS, cS, b.memP + oP, cP, mode, lane_lds + (threadIdx.x & 63) * KC_LANE_STRIDE, true, g.tag_base + (u32)t);
            continue;
        }
        ovg_insertion_sort(b.memS + oS, cS);
        ovg_insertion_sort(b.memP + oP, cP);
        kc_match_bucket(g, b.memS + oS, cS, b.memP + oP, cP, mode, g.tag_base + (u32)t, b.scrS + oS, b.scrP + oP);
    }
}
// one wavefront per listed bucket (workgroups of 64 take the descriptors in turn)
__global__ void __launch_bounds__(64) ovg_match_lds_kernel(OvArgs g, const u64 *__restrict__ st, OvgBuf b, int huge, u32 lo, u32 hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char kc_lds[];
    if (st[OVG_STATUS]) return;
    u64 nb = st[huge ? OVG_HUGE : OVG_BIG];
    const u64 cap = huge ? b.huge_cap : b.big_cap;
    if (nb > cap) nb = cap;
    const u64 *list = huge ? b.huge : b.big;
    for (u64 i = blockIdx.x; i < nb; i += gridDim.x) {
        const u64 *dsc = list + i * 5;
        const u64 members = (dsc[1] - dsc[0]) + (dsc[3] - dsc[2]);
        if (members <= lo || members > hi) continue;   // (another launch's size class: see ov_match_lds_kernel)
        kc_match_bucket_lds(g, b.memS + dsc[0], (u32)(dsc[1] - dsc[0]), b.memP + dsc[2], (u32)(dsc[3] - dsc[2]), (int)dsc[4], kc_lds, true);
        __syncthreads();
    }
}
// verification: as ov_fix_kernel<false>, over the gathered log, any grid
__global__ void __launch_bounds__(256) ovg_cover_kernel(u64 *st, const u64 *__restrict__ ne, const u64 *__restrict__ n_ptr, const u64 *__restrict__ edge_from,
                                                        const u64 *__restrict__ pred, const u64 *__restrict__ first_snap, const u64 *__restrict__ last_snap, u8 *cov) {
    if (st[OVG_STATUS]) return;
    const u64 n = *n_ptr;
    u64 mine = 0;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (u64)gridDim.x * blockDim.x) {
        const u64 y = ne[t * 5 + 1];
        const u64 l0 = last_snap[y];
        if (edge_from[l0] != KC_NONE64) continue;   // not the tail of its merged path
        u64 f = y, steps = 0;
        while (steps <= n) {
            const u64 xp = pred[f];
            if (xp == KC_NONE64) break;
            cov[f] = 1;
            ++steps;
            f = first_snap[xp];
        }
        mine += steps;
    }
    if (mine) atomicAdd((ull *)&st[OVG_COVERED], (ull)mine);
}
// new path ends + coverage marks cleared (ov_fix_write_clear_kernel, any grid) -- and the level's verdict and book-keeping: the level is
// acyclic iff every new edge was covered by a walk from a path's tail (else the host repairs it); every new edge removes exactly one
// suffix vertex and one prefix vertex
__global__ void __launch_bounds__(256) ovg_write_kernel(u64 *st, const u64 *__restrict__ ne, const u64 *__restrict__ n_ptr, const u64 *__restrict__ shflag,
                                                        const u64 *__restrict__ edge_from, const u64 *__restrict__ pred, u64 *first_snap, u64 *last_snap, u8 *cov, int d) {
    if (st[OVG_STATUS]) return;
    const u64 n = *n_ptr;
    const bool log_lost = *shflag != 0, cyclic = st[OVG_COVERED] != n;   // (the same for every thread: nothing in this kernel writes these words)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st[OVG_NE] = n;
        if (log_lost) ovg_poison(st, d, OVG_REASON_LOG);
        else if (cyclic) ovg_poison(st, d, OVG_REASON_REPAIR);
        else {
            st[OVG_LIVES] -= n; st[OVG_LIVEP] -= n; st[OVG_EDGES] += n;
            st[OVG_LEVELS_DONE] = (u64)d;
            st[OVG_NEWS] = 0; st[OVG_NEWP] = 0;
            st[OVG_COMPACT] = ((st[OVG_LENS] - st[OVG_LIVES]) * 4 > st[OVG_LENS] || (st[OVG_LENP] - st[OVG_LIVEP]) * 4 > st[OVG_LENP]) ? 1 : 0;
        }
    }
    if (log_lost || cyclic) return;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (u64)gridDim.x * blockDim.x) {
        const u64 y = ne[t * 5 + 1];
        cov[y] = 0;
        const u64 l0 = last_snap[y];
        if (edge_from[l0] != KC_NONE64) continue;
        u64 f = y, steps = 0;
        while (steps <= n) {
            const u64 xp = pred[f];
            if (xp == KC_NONE64) break;
            ++steps;
            f = first_snap[xp];
        }
        first_snap[l0] = f; last_snap[f] = l0;
    }
}
// lazy compaction of the active lists (the order inside them does not matter to the grouped levels)
__global__ void __launch_bounds__(256) ovg_compact_kernel(u64 *st, OvgBuf b, const u64 *__restrict__ edge_from, const u64 *__restrict__ pred) {
    if (st[OVG_STATUS] || !st[OVG_COMPACT]) return;
    const u64 nS = st[OVG_LENS], nP = st[OVG_LENP];
    const int cur = (int)(st[OVG_CUR] & 1);
    const u64 *srcS = b.act[cur], *srcP = b.act[2 + cur];
    u64 *dstS = b.act[cur ^ 1], *dstP = b.act[2 + (cur ^ 1)];
    for (u64 i0 = (u64)blockIdx.x * blockDim.x; i0 < nS + nP; i0 += (u64)gridDim.x * blockDim.x) {
        const u64 i = i0 + threadIdx.x;
        u64 v = 0;
        bool keepS = false, keepP = false;
        if (i < nS) { v = srcS[i]; keepS = edge_from[v] == KC_NONE64; }
        else if (i < nS + nP) { v = srcP[i - nS]; keepP = pred[v] == KC_NONE64; }
        const u64 pS = kc_block_reserve(keepS ? 1u : 0u, &st[OVG_NEWS]);
        const u64 pP = kc_block_reserve(keepP ? 1u : 0u, &st[OVG_NEWP]);
        if (keepS) dstS[pS] = v;
        if (keepP) dstP[pP] = v;
    }
}
// ---------------------------------------------------------------------------------------------------------
struct OvWork;
static int read_back(kc_ctx *ctx, u64 *h_pinned, const u64 *d, int n) {
    KC_HIP(ctx, hipMemcpyAsync(h_pinned, d, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}
struct OvWork {
    kc_ctx *ctx;
    void *tmp = nullptr; size_t tmp_bytes = 0;
    u64 *d_num = nullptr;
    u64 *d_flag = nullptr, *h_flag = nullptr;   // bucketed sort: overflow flag (device word, pinned host word)
    int ensure_tmp(size_t bytes) {
        if (bytes <= tmp_bytes) return KC_OK;
        kc_release(ctx, tmp);
        tmp = nullptr; tmp_bytes = 0;
        u8 *p = nullptr;
        KC_TRY(kc_alloc(ctx, &p, bytes));
        tmp = p; tmp_bytes = bytes;
        return KC_OK;
    }
};

// ordered (stable) compaction by wave ballot / prefix scan: kcu_select_flagged (kc_util.hip)
static int select_flagged(OvWork &w, const u64 *in, const u8 *flags, u64 *out, u64 n, u64 *h_count) {
    kc_ctx *ctx = w.ctx;
    KC_TRY(kcu_select_flagged(ctx, in, flags, n, out, w.d_num, false));
    KC_HIP(ctx, hipMemcpyAsync(h_count, w.d_num, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}

static int select_flagged_nosync(OvWork &w, const u64 *in, const u8 *flags, u64 *out, u64 n) {
    return kcu_select_flagged(w.ctx, in, flags, n, out, w.d_num, false);
}

// Small inputs (most levels: a few hundred to a few thousand hits) are sorted by ONE workgroup in LDS -- a bitonic network
// over (key, id) rows; ids are ascending in the input, so ordering by (key, id) is the stable order by key.  The library
// radix sort costs 4-8 launches per call, which is what a level with few hits was spending its time on.
// MODE 0: keys given; MODE 1 / 2: the key of row i is the d-mer suffix / prefix of ends[ids[i]] (the rows may then come in
// any order, e.g. straight from the hit list of the probe kernels).
#define KC_SMALL_ROWS 1024
template <int W, int MODE>
__global__ void __launch_bounds__(1024) ov_small_sort_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ ids, u32 n, u32 P,
                                                             const u64 *__restrict__ ends, int k, int d, u64 *keys_out, u64 *ids_out) {
    extern __shared__ u64 kc_srt[];   // P ids, then W planes of P key words
    u64 *sid = kc_srt;
    u64 *sk = kc_srt + P;
    for (u32 i = threadIdx.x; i < P; i += blockDim.x) {
        const u64 id = i < n ? ids[i] : ~(u64)0;
        sid[i] = id;
        Key<W> key;
        for (int j = 0; j < W; ++j) key.w[j] = ~(u64)0;
        if (i < n) {
            if (MODE == 0) key = load_key<W>(keys, i);
            else { Key<W> e = load_key<W>(ends, id); key = (MODE == 1) ? key_suffix(e, d) : key_prefix(e, k, d); }
        }
        for (int j = 0; j < W; ++j) sk[(u32)j * P + i] = key.w[j];
    }
    __syncthreads();
    auto greater = [&](u32 a, u32 b) -> bool {   // row a sorts after row b
        for (int j = W - 1; j >= 0; --j) {
            const u64 x = sk[(u32)j * P + a], y = sk[(u32)j * P + b];
            if (x != y) return x > y;
        }
        return sid[a] > sid[b];
    };
    for (u32 size = 2; size <= P; size <<= 1) {
        for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
            for (u32 t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const u32 lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                if (greater(lo, hi) == up) {
                    u64 tmp = sid[lo]; sid[lo] = sid[hi]; sid[hi] = tmp;
                    for (int j = 0; j < W; ++j) { tmp = sk[(u32)j * P + lo]; sk[(u32)j * P + lo] = sk[(u32)j * P + hi]; sk[(u32)j * P + hi] = tmp; }
                }
            }
            __syncthreads();
        }
    }
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        ids_out[i] = sid[i];
        for (int j = 0; j < W; ++j) keys_out[(u64)i * W + j] = sk[(u32)j * P + i];
    }
}

// stable sort of n rows (W-word keys, 2*d significant bits) with their ids; outputs in keys_out / ids_out
// side (0 suffix / 1 prefix) selects the overflow word of the bucketed sort; *bucketed tells the caller that the result is only
// valid if that word is still zero (it is read back together with the match counters: no sync of its own);
// allow_bucketed = false: the LSD sort right away
static int sort_rows(OvWork &w, const u64 *keys, const u64 *ids, u64 n, int W, int d, u64 *keys_out, u64 *ids_out, u64 *s0, u64 *s1,
                     u64 *s2, u64 *s3, int side, bool allow_bucketed, bool *bucketed) {
    *bucketed = false;
    kc_ctx *ctx = w.ctx;
    int total_bits = 2 * d;
    if (total_bits == 0) {
        KC_HIP(ctx, hipMemcpyAsync(keys_out, keys, n * W * 8, hipMemcpyDeviceToDevice, ctx->stream));
        KC_HIP(ctx, hipMemcpyAsync(ids_out, ids, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return KC_OK;
    }
    if (n <= KC_SMALL_ROWS && !getenv("KC_NO_SMALL_SORT")) {   // beyond ~1k rows the library radix sort wins (measured)
        u32 P = 2;
        while (P < n) P <<= 1;
        const size_t lds = (size_t)P * (W + 1) * 8;
        switch (W) {
            case 1: KC_LAUNCH((ov_small_sort_kernel<1, 0>), dim3(1), dim3(1024), lds, ctx->stream, keys, ids, (u32)n, P, (const u64 *)nullptr, 0, d, keys_out, ids_out); break;
            case 2: KC_LAUNCH((ov_small_sort_kernel<2, 0>), dim3(1), dim3(1024), lds, ctx->stream, keys, ids, (u32)n, P, (const u64 *)nullptr, 0, d, keys_out, ids_out); break;
            default: KC_LAUNCH((ov_small_sort_kernel<4, 0>), dim3(1), dim3(1024), lds, ctx->stream, keys, ids, (u32)n, P, (const u64 *)nullptr, 0, d, keys_out, ids_out); break;
        }
        KC_HIP(ctx, hipGetLastError());
        return KC_OK;
    }
    // mid-size and large lists: one MSD bucket pass + one-workgroup LDS sorts (4 launches); skewed keys overflow a bucket and
    // take the LSD radix sort below (5 launches per 4 key bits) instead
    if (allow_bucketed && !getenv("KC_NO_BUCKET_SORT")) {
        const int brc = kcu_sort_rows_bucketed(ctx, keys, ids, n, W, total_bits, keys_out, ids_out, w.d_flag + side);
        if (brc == KC_OK) { *bucketed = true; return KC_OK; }
        if (brc != KC_E_ARG) return brc;
    }
    if (W == 1) return kcu_sort_pairs(ctx, keys, keys_out, ids, ids_out, n, 0, total_bits, false);
    u64 *perm = s0, *perm2 = s1, *kw = s2, *kw2 = s3;
    KC_LAUNCH(ov_iota_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, perm, n);
    for (int word = 0; word < W; ++word) {
        int bits = total_bits - 64 * word;
        if (bits <= 0) break;
        if (bits > 64) bits = 64;
        KC_LAUNCH(ov_gather_word_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, keys, (const u64 *)perm, n, W, word, kw);
        KC_TRY(kcu_sort_pairs(ctx, kw, kw2, perm, perm2, n, 0, bits, false));
        std::swap(perm, perm2);
    }
    KC_LAUNCH(ov_gather_rows_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, keys, ids, (const u64 *)perm, n, W, keys_out,
                       ids_out);
    KC_HIP(ctx, hipGetLastError());
    return KC_OK;
}

template <int W> static int overlap_path_w(kc_simplitigs *s, int complements, int lower_bound, KcPathDev *out, kc_comm *comm) {
    kc_ctx *ctx = s->ctx;
    KcScope scope(ctx);
    // Hash-sharded flow (comm with more than one rank): every rank holds all vertices, but the KEYS of a level are divided by
    // hash -- a rank builds the level table, finds the hits and replays the buckets of its keys only, then the ranks
    // all-gather the edges they logged and apply the others'; verification, repair and the shrinking of the active lists run
    // on the combined log, identically everywhere.  Exactness is untouched: buckets of one level never shared state except
    // through the verified path ends (that is what the verify / repair step is for), whether they run on one GPU or on eight.
    // WHICH levels the ranks divide (round 5): only the levels that build a table of their own while more than 2^20 vertices are live
    // (KC_OVG_MAX) -- the mid levels of a large read set, where a level costs sweeps, sorts and hundreds of thousands of small replays, all
    // of which divide by the number of ranks.  Every other level runs WHOLE on every rank, through the single-GPU path and without a
    // collective: the levels d >= 16 through the 16-mer table (one cheap sweep each, few hits), and everything from 2^20 live vertices
    // down (the grouped levels: bound by launches, and at d <= 3 by a few huge buckets that one wavefront replays -- nothing to divide).
    // All ranks hold identical vertex arrays after every level (the divided ones all-gather and apply their edges), and what decides the
    // mode -- d, the live counts, whether the 16-mer table is in use -- is replicated, so the ranks switch together.  Measured with 8 ranks
    // on 10^8 k-mers (2 10^5 vertices): every level divided 20 ms and 52 collectives, against 5 ms for the phase on one GPU; at cfg[3]
    // (5 10^6 vertices) the whole phase is 27 ms on one GPU, of which the 8 divided levels d = 16 .. 9 are 15.
    // KC_SHARD_REPLICATE_PHASE2=1: never divide; KC_SHARD_SPLIT_MIN=v: only from v vertices on (default 0).
    const int k = s->k;
    const u64 S = s->n, V = S * (complements ? 2 : 1);
    u64 split_min = 0;
    if (const char *e = getenv("KC_SHARD_SPLIT_MIN")) split_min = (u64)atoll(e);
    const bool can_split = comm && comm->world > 1 && !lower_bound && !getenv("KC_SHARD_REPLICATE_PHASE2") && V >= split_min;
    bool split = false;      // the level at hand is divided (set per level in the loop at the end)
    u32 PG = 1u, pme = 0u;
    if (S == 0) return kc_fail_agreed(ctx, KC_E_EMPTY, "input cannot be empty");
    if (V >= ((u64)1 << 32)) return kc_fail(ctx, KC_E_ARG, "more than 2^32 overlap-graph vertices are not supported (32-bit writer tags of the path-end entries)");
    OvWork w; w.ctx = ctx;
    u64 *pre, *suf, *edge_from, *pred, *first, *last, *first_snap, *last_snap, *actS, *actP, *actS2, *actP2, *eidx;
    u8 *overlaps, *hitS, *hitP, *cov;
    u64 *pslot, *ne, *cnt, *cnt_pool, *bl;
    KC_TRY(scope.alloc(&pre, V * W)); KC_TRY(scope.alloc(&suf, V * W));
    edge_from = pred = nullptr; overlaps = nullptr;
    u64 *hp = nullptr;   // pinned readback buffer
    bool ok = false;
    auto cleanup = kc_defer([&] {   // on failure the three output arrays go back too
        if (hp) (void)hipHostFree(hp);
        if (!ok) { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, edge_from); kc_release(ctx, pred); kc_release(ctx, overlaps); }
    });
    KC_TRY(kc_alloc(ctx, &edge_from, V)); KC_TRY(kc_alloc(ctx, &pred, V)); KC_TRY(kc_alloc(ctx, &overlaps, V));
    KC_TRY(scope.alloc(&first, V)); KC_TRY(scope.alloc(&last, V)); KC_TRY(scope.alloc(&first_snap, V)); KC_TRY(scope.alloc(&last_snap, V));
    KC_TRY(scope.alloc(&actS, V)); KC_TRY(scope.alloc(&actP, V)); KC_TRY(scope.alloc(&actS2, V)); KC_TRY(scope.alloc(&actP2, V));
    KC_TRY(scope.alloc(&eidx, V)); KC_TRY(scope.alloc(&hitS, V)); KC_TRY(scope.alloc(&hitP, V)); KC_TRY(scope.alloc(&cov, V));
    KC_TRY(scope.alloc(&pslot, V)); KC_TRY(scope.alloc(&ne, V * 10 + 10)); KC_TRY(scope.alloc(&cnt_pool, KC_CNT_BLOCKS * 16));  // ne: up to two log entries per event
    // counter blocks (16 words): every use takes a fresh, already zeroed block instead of clearing one with a fill launch
    KC_HIP(ctx, hipMemsetAsync(cnt_pool, 0, KC_CNT_BLOCKS * 16 * 8, ctx->stream));
    u64 cnt_used = 0;
    auto fresh_cnt = [&]() -> u64 * {
        if (cnt_used == KC_CNT_BLOCKS) { (void)hipMemsetAsync(cnt_pool, 0, KC_CNT_BLOCKS * 16 * 8, ctx->stream); cnt_used = 0; }
        return cnt_pool + 16 * cnt_used++;
    };
    cnt = fresh_cnt();
    const u64 BL_MAX = 4096;
    KC_TRY(scope.alloc(&bl, BL_MAX * 2));
    KC_TRY(scope.alloc(&w.d_num, 4));
    w.d_flag = w.d_num + 2;   // two words: overflow of the suffix-side / prefix-side bucketed sort of the current level
    u64 *big = nullptr;
    int lds_min = KC_LDS_MIN;
    if (const char *e = getenv("KC_LDS_MIN")) { int v = atoi(e); if (v >= 2 && v <= KC_LDS_MAX) lds_min = v; }
    const u64 big_cap = 2 * V / (u64)(lds_min < 4 ? lds_min : 4) + 16;
    KC_TRY(scope.alloc(&big, big_cap * 5));
    u64 *huge = nullptr;
    const u64 huge_cap = 2 * V / KC_LDS_MAX + 16;
    KC_TRY(scope.alloc(&huge, huge_cap * 5));
    KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ov_match_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kc_lds_bytes_bound(KC_LDS_HUGE)));
    u64 tcap = 64;
    while (tcap < 2 * V + 16) tcap <<= 1;
    LevelTable<W> tab;
    KC_TRY(scope.alloc(&tab.slot, tcap + tcap / 2 + 1));   // slots (8 B) directly followed by the flags (4 B): one fill per level
    tab.flags = nullptr;
    // hit buffers
    u64 *hs_ids, *hp_ids, *hs_keys, *hp_keys, *hs_ids2, *hp_ids2, *hs_keys2, *hp_keys2, *sc0, *sc1, *sc2, *sc3;
    KC_TRY(scope.alloc(&hs_ids, V)); KC_TRY(scope.alloc(&hp_ids, V)); KC_TRY(scope.alloc(&hs_ids2, V)); KC_TRY(scope.alloc(&hp_ids2, V));
    KC_TRY(scope.alloc(&hs_keys, V * W)); KC_TRY(scope.alloc(&hp_keys, V * W)); KC_TRY(scope.alloc(&hs_keys2, V * W)); KC_TRY(scope.alloc(&hp_keys2, V * W));
    sc0 = sc1 = sc2 = sc3 = nullptr;
    if (W > 1) { KC_TRY(scope.alloc(&sc0, V)); KC_TRY(scope.alloc(&sc1, V)); KC_TRY(scope.alloc(&sc2, V)); KC_TRY(scope.alloc(&sc3, V)); }

    KC_HIP(ctx, hipMemsetAsync(cov, 0, V, ctx->stream));
    KC_LAUNCH((ov_init_kernel<W>), dim3(kc_grid(V, 256)), dim3(256), 0, ctx->stream, (const u64 *)s->firstK, (const u64 *)s->lastK, S, V, k,
                       pre, suf, edge_from, overlaps, pred, first, last, first_snap, last_snap, actS, actP, eidx);
    KC_HIP(ctx, hipGetLastError());

    // the 16-mer table of the prefix ends for the levels d >= 16 (large inputs; KC_T16_MIN = vertex count from which it is used)
    T16 t16{};
    u32 *markP = nullptr;
    u64 t16_min = (u64)1 << 16;
    if (const char *e = getenv("KC_T16_MIN")) { const long long v = atoll(e); if (v >= 0) t16_min = (u64)v; }
    // (simplitigs of reads with coverage -- the set build saw every k-mer three times or more -- overlap at every level: the per-level
    // tables are the faster path there, see the sampled check below; the table is not even built)
    const bool reads_with_coverage = s->occ_ratio >= 3.0f && !getenv("KC_T16_FORCE");
    const bool t16_on = k - 1 >= 16 && V >= t16_min && V < ((u64)1 << 32) - 1 && !reads_with_coverage;
    bool t16_dead = false;   // the table is not (or no longer) worth using
    if (getenv("KC_OV_TRACE")) fprintf(stderr, "[ov] V=%llu, occurrences per distinct k-mer of the set %.3f -> 16-mer table %s\n", (unsigned long long)V, (double)s->occ_ratio, t16_on ? "on" : "off");
    if (t16_on) {
        u64 c16 = 64;
        while (c16 < 2 * V + 16) c16 <<= 1;
        t16.mask = c16 - 1;
        KC_TRY(scope.alloc(&t16.key, c16)); KC_TRY(scope.alloc(&t16.head, c16)); KC_TRY(scope.alloc(&t16.next, V)); KC_TRY(scope.alloc(&markP, V));
        KC_HIP(ctx, hipMemsetAsync(t16.key, 0xff, c16 * 8, ctx->stream));
        KC_HIP(ctx, hipMemsetAsync(t16.head, 0xff, c16 * 4, ctx->stream));
        KC_HIP(ctx, hipMemsetAsync(markP, 0, V * 4, ctx->stream));
        u64 *dcnt = fresh_cnt();
        KC_LAUNCH((ov_t16_build_kernel<W>), dim3(kc_grid(V, 256)), dim3(256), 0, ctx->stream, t16, (const u64 *)pre, V, k, dcnt);
        KC_HIP(ctx, hipGetLastError());
        // ends that share their first 16 bases (coverage: many simplitigs begin inside the same stretch of genome) make every
        // lookup compare several candidates exactly; measured, the per-level tables win from ~1.3 vertices per 16-mer on
        u64 distinct = 0;
        KC_HIP(ctx, hipMemcpyAsync(&distinct, dcnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (distinct * 13 < V * 10 && !getenv("KC_T16_FORCE")) t16_dead = true;
        if (getenv("KC_OV_TRACE")) fprintf(stderr, "[ov] 16-mer table: %llu vertices, %llu distinct 16-mers -> %s\n", (unsigned long long)V, (unsigned long long)distinct, t16_dead ? "not used" : "used");
    }
    OvArgs g;
    g.S = S; g.V = V; g.batch = V / 16 + 1; g.complements = complements; g.lower_bound = lower_bound;
    g.edge_from = edge_from; g.overlaps = overlaps; g.pred = pred; g.first = first; g.last = last;
    g.first_snap = first_snap; g.last_snap = last_snap;
    g.ne = ne; g.ne_count = cnt; g.bl = bl; g.nbl = 0; g.lds_min = (u32)lds_min; g.abort2 = nullptr;
    g.lane_off = getenv("KC_OV_NO_LANE_REPLAY") ? 1u : 0u;   // (test knob: small buckets replayed from global memory, one thread each -- the path of buckets between KC_LANE_MAX and lds_min members)
#ifdef KC_EXPERIMENTS
    if (const char *e = KC_EXP_KNOB("KC_MATCH_EXP")) g.exp = (u32)atoi(e);
#endif
    // sharded new-edge log (see ov_add_event): every shard could hold a whole level's edges of an evenly spread load four times over
    u64 *ne_sh = nullptr, *ne_shcur = nullptr;
    const u64 ne_shcap = (2 * V) / 16 + 1024;
    const bool log_sharded = (V >= ((u64)1 << 16) && !getenv("KC_NO_SHARDED_LOG")) || (!lower_bound && !getenv("KC_OV_HOST_LOOP"));   // (the grouped levels always log through the shards)
    if (log_sharded) { KC_TRY(scope.alloc(&ne_sh, 64 * ne_shcap * 5)); KC_TRY(scope.alloc(&ne_shcur, 65 * 16)); }
    g.ne_sh = nullptr; g.ne_shcur = ne_shcur; g.ne_shcap = ne_shcap;
    u64 nS = V, nP = V, repairs = 0, tag_base = 0;
    u64 lenS = V, lenP = V;   // entries of the active lists (>= the live vertices nS / nP: the lists are compacted lazily, see the end of the loop)
    // windowed hit counts through the 16-mer table (ov_count16_kernel): levels [win_lo, win_hi] have been counted
    u64 *whist = nullptr;
    KC_TRY(scope.alloc(&whist, 64));
    int win_hi = -1, win_lo = 0;
    u64 wcount[32] = {0};
    bool win_on = t16_on && !getenv("KC_OV_NO_WINDOW");   // (the levels through the 16-mer table are never divided among ranks)
    bool t16_sampled = false;
    std::vector<u64> h_bl;
    if (!lower_bound) {   // tag 0xffffffff = written by nobody
        KC_HIP(ctx, hipMemsetAsync(first, 0xff, V * 8, ctx->stream));
        KC_HIP(ctx, hipMemsetAsync(last, 0xff, V * 8, ctx->stream));
    }

    KC_HIP(ctx, hipHostMalloc((void **)&hp, 1024, hipHostMallocDefault));   // (hp[64..96): the window counts)
    w.h_flag = hp + 15;   // (the level loop reads back at most 9 words; hp[32..] are staging words of the sharded flow)
    u64 *ne_all = nullptr;
    if (can_split) KC_TRY(scope.alloc(&ne_all, V * 10 + 10));
    auto run_level_host = [&](const int d) -> int {   // one level of the host-driven loop
        g.d = d;
        const bool trace = getenv("KC_OV_TRACE") != nullptr;
        auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double t_level = 0, t_hits = 0, t_sorted = 0, t_matched = 0;
        if (trace) { (void)hipStreamSynchronize(ctx->stream); t_level = now_us(); }
        // 1. bucket: cnt[1] = suffix hits, cnt[3] = prefix hits
        auto own_table_probe = [&]() -> int {   // the level builds a table of its own prefix keys
            u64 lcap = 64;
            const u64 tab_keys = split ? nP / PG + nP / (4 * PG) + 512 : nP;   // (our share of the keys, with room for an uneven split)
            while (lcap < 2 * tab_keys + 16) lcap <<= 1;
            tab.mask = lcap - 1;
            tab.flags = reinterpret_cast<u32 *>(tab.slot + lcap);
            KC_HIP(ctx, hipMemsetAsync(tab.slot, 0xff, lcap * 12, ctx->stream));
            cnt = fresh_cnt();
            KC_LAUNCH((ov_insert_prefix_kernel<W>), dim3(kc_grid(lenP, 256)), dim3(256), 0, ctx->stream, tab, (const u64 *)actP, lenP,
                               (const u64 *)pre, (const u64 *)pred, k, d, pslot, PG, pme);
            KC_LAUNCH((ov_probe_suffix_kernel<W>), dim3(kc_grid((lenS + OV_ITEMS - 1) / OV_ITEMS, 256)), dim3(256), 0, ctx->stream, tab, (const u64 *)actS, lenS,
                               (const u64 *)suf, (const u64 *)pre, (const u64 *)edge_from, k, d, complements, hitS, cnt + 1, hs_ids, KC_SMALL_ROWS, PG, pme);
            KC_LAUNCH((ov_flag_prefix_kernel<W>), dim3(kc_grid((lenP + OV_ITEMS - 1) / OV_ITEMS, 256)), dim3(256), 0, ctx->stream, tab, (const u64 *)actP, lenP,
                               (const u64 *)pre, (const u64 *)pslot, k, d, complements, hitP, cnt + 3, hp_ids, KC_SMALL_ROWS);
            KC_TRY(read_back(ctx, hp, cnt, 4));
            return KC_OK;
        };
        if (t16_on && !t16_dead && d >= 16 && !t16_sampled && !split && lenS >= ((u64)1 << 20) && !getenv("KC_T16_FORCE")) {
            // Many vertices: are these overlapping reads?  The hits of a sample of the suffix ends at this level and the next ones
            // tell -- with a hit for every 64th vertex the per-level tables are the faster path (below), and finding that out by
            // running the first level through the 16-mer table cost 14 ms on 3.9 10^7 vertices.
            t16_sampled = true;
            const u64 ns = (u64)1 << 18;
            const int B = d - 15 < 4 ? d - 15 : 4;
            KC_HIP(ctx, hipMemsetAsync(whist, 0, 32 * 8, ctx->stream));
            KC_LAUNCH((ov_count16_kernel<W>), dim3(kc_grid(ns, 256)), dim3(256), 0, ctx->stream, t16, (const u64 *)actS, ns, (const u64 *)suf,
                               (const u64 *)pre, (const u64 *)pred, (const u64 *)edge_from, k, d, B, whist, PG, pme);
            KC_TRY(read_back(ctx, hp + 64, whist, B));
            u64 sh = 0;
            for (int j = 0; j < B; ++j) sh += hp[64 + j];
            if (sh * 64 > ns * (u64)B) t16_dead = true;
            if (trace) fprintf(stderr, "[ov] sample of %llu suffix ends, levels %d..%d: %llu hits -> 16-mer table %s\n", (unsigned long long)ns, d, d - B + 1, (unsigned long long)sh, t16_dead ? "not used" : "used");
        }
        if (t16_on && !t16_dead && d >= 16) {
            // Small active sets: count the hits of a window of levels in one sweep first (ov_count16_kernel); a level that counts
            // zero is skipped without a launch or a read-back.  The window shrinks to nothing once most levels have hits.
            if (win_on && (d > win_hi || d < win_lo)) {
                int B = (int)(((u64)1 << 22) / (lenS ? lenS : 1));
                if (B > 32) B = 32;
                if (B > d - 15) B = d - 15;
                if (B >= 4) {   // (a window of two or three levels costs what the levels cost)
                    KC_HIP(ctx, hipMemsetAsync(whist, 0, 32 * 8, ctx->stream));
                    KC_LAUNCH((ov_count16_kernel<W>), dim3(kc_grid(lenS, 256)), dim3(256), 0, ctx->stream, t16, (const u64 *)actS, lenS, (const u64 *)suf,
                                       (const u64 *)pre, (const u64 *)pred, (const u64 *)edge_from, k, d, B, whist, PG, pme);
                    KC_TRY(read_back(ctx, hp + 64, whist, B));
                    int with_hits = 0;
                    for (int j = 0; j < B; ++j) { wcount[j] = hp[64 + j]; with_hits += wcount[j] != 0; }
                    win_hi = d; win_lo = d - B + 1;
                    if (2 * with_hits > B) win_on = false;   // (hits at most levels: the count costs as much as the levels themselves)
                    if (trace) fprintf(stderr, "[ov] window d=%d..%d: %d level(s) with hits\n", win_hi, win_lo, with_hits);
                } else win_on = false;
            }
            if (d <= win_hi && d >= win_lo && wcount[win_hi - d] == 0) return KC_OK;   // counted: no suffix vertex can hit at this level
            // one sweep over the active suffixes through the 16-mer table of the prefix ends (see ov_probe16_kernel)
            cnt = fresh_cnt();
            KC_LAUNCH((ov_probe16_kernel<W>), dim3(kc_grid(lenS, 256)), dim3(256), 0, ctx->stream, t16, (const u64 *)actS, lenS, (const u64 *)suf,
                               (const u64 *)pre, (const u64 *)pred, (const u64 *)edge_from, k, d, complements, hitS, markP, cnt, hs_ids, hp_ids, (u64)KC_SMALL_ROWS, PG, pme);
            KC_TRY(read_back(ctx, hp, cnt, 6));
            if (hp[4]) {   // a chain beyond the limit (repetitive ends; chains never shrink): this level and the later ones take tables of their own
                t16_dead = true;
                KC_HIP(ctx, hipMemsetAsync(markP, 0, V * 4, ctx->stream));
                KC_TRY(own_table_probe());
            } else if (hp[3]) {
                if (hp[5] || hp[3] >= KC_SMALL_ROWS || getenv("KC_NO_SMALL_SORT")) {
                    // many prefix hits: their exact number and hitP (read by the ordered compaction) come from a sweep over the marks
                    u64 *c2 = fresh_cnt();
                    KC_LAUNCH(ov_marks_to_hits_kernel, dim3(kc_grid(lenP, 256)), dim3(256), 0, ctx->stream, (const u64 *)actP, lenP, (const u32 *)markP, hitP,
                                       c2, hp_ids, (u64)KC_SMALL_ROWS);
                    KC_TRY(read_back(ctx, hp + 48, c2, 1));
                    hp[3] = hp[48];
                }
                KC_HIP(ctx, hipMemsetAsync(markP, 0, V * 4, ctx->stream));
                // many hits at a high level (overlapping reads): every hit costs three scattered accesses here, and measured the
                // per-level tables are ~3x faster on such data -- the later levels use them
                if (hp[3] > V / 64 && !getenv("KC_T16_FORCE")) t16_dead = true;
            }
        } else KC_TRY(own_table_probe());
        const u64 nHS = hp[1], nHP = hp[3];
        const bool local_any = !(nHS == 0 || nHP == 0);
        // (sharded flow: whether any rank has hits shows in the all-gather of the edge counts below -- one collective per level)
        const bool any = local_any || split;
        if (trace) t_hits = now_us();
        if (!any && trace) fprintf(stderr, "[ov] d=%d nS=%llu no hits: %.0f us\n", d, (unsigned long long)nS, t_hits - t_level);
        if (!any) return KC_OK;   // no key seen from both sides at this overlap length
        // 2. compact the hits (ordered; the counts are already known) ...
        // 3. ... and sort them by key (stable: ids ascending inside a key).  Few hits: the probe kernels have listed them
        // already (unordered); one workgroup derives the keys and sorts the rows by (key, id) in LDS.
        const bool small_ok = !getenv("KC_NO_SMALL_SORT");
        auto small_side = [&](bool suffix_side, const u64 *ids_in, u64 n, u64 *keys_o, u64 *ids_o) -> int {
            u32 P = 2;
            while (P < n) P <<= 1;
            const size_t lds = (size_t)P * (W + 1) * 8;
            if (suffix_side)
                KC_LAUNCH((ov_small_sort_kernel<W, 1>), dim3(1), dim3(1024), lds, ctx->stream, (const u64 *)nullptr, ids_in, (u32)n, P,
                                   (const u64 *)suf, k, d, keys_o, ids_o);
            else
                KC_LAUNCH((ov_small_sort_kernel<W, 2>), dim3(1), dim3(1024), lds, ctx->stream, (const u64 *)nullptr, ids_in, (u32)n, P,
                                   (const u64 *)pre, k, d, keys_o, ids_o);
            return KC_OK;
        };
        // a bucketed sort that overflows (skewed keys) leaves its flag word set and its output void; the match kernels then
        // stand down, the flag comes back with their counters, and the side is sorted again by the LSD sort (no sync of its own)
        bool bkS = false, bkP = false;
        KC_HIP(ctx, hipMemsetAsync(w.d_flag, 0, 16, ctx->stream));
        if (!local_any) { /* nothing of ours at this level: only the exchange below */ }
        else if (small_ok && nHS <= KC_SMALL_ROWS) KC_TRY(small_side(true, hs_ids, nHS, hs_keys2, hs_ids2));
        else {
            KC_TRY(select_flagged_nosync(w, actS, hitS, hs_ids, lenS));
            KC_LAUNCH((ov_hit_keys_kernel<W, true>), dim3(kc_grid(nHS, 256)), dim3(256), 0, ctx->stream, (const u64 *)hs_ids, nHS,
                               (const u64 *)suf, k, d, hs_keys);
            KC_TRY(sort_rows(w, hs_keys, hs_ids, nHS, W, d, hs_keys2, hs_ids2, sc0, sc1, sc2, sc3, 0, true, &bkS));
        }
        if (!local_any) { }
        else if (small_ok && nHP <= KC_SMALL_ROWS) KC_TRY(small_side(false, hp_ids, nHP, hp_keys2, hp_ids2));
        else {
            KC_TRY(select_flagged_nosync(w, actP, hitP, hp_ids, lenP));
            KC_LAUNCH((ov_hit_keys_kernel<W, false>), dim3(kc_grid(nHP, 256)), dim3(256), 0, ctx->stream, (const u64 *)hp_ids, nHP,
                               (const u64 *)pre, k, d, hp_keys);
            KC_TRY(sort_rows(w, hp_keys, hp_ids, nHP, W, d, hp_keys2, hp_ids2, sc0, sc1, sc2, sc3, 1, true, &bkP));
        }
        KC_HIP(ctx, hipGetLastError());
        if (trace) { (void)hipStreamSynchronize(ctx->stream); t_sorted = now_us(); }
        // 4./5. match, verify, repair
        u64 nNew = 0;
        const u64 max_new = (complements ? 2 : 1) * nHS;   // every event logs at most 1 (+1 mirrored) edge
        while (true) {
            cnt = fresh_cnt();
            g.ne_count = cnt;
            if (!lower_bound) {
                // path-end entries carry the tag of the bucket that wrote them; every launch draws its tags from a fresh
                // range, so entries of other levels / replays are simply never trusted (no clearing between launches)
                if (tag_base + nHS >= 0xfffffff0ull) {
                    KC_HIP(ctx, hipMemsetAsync(first, 0xff, V * 8, ctx->stream));
                    KC_HIP(ctx, hipMemsetAsync(last, 0xff, V * 8, ctx->stream));
                    tag_base = 0;
                }
                g.tag_base = (u32)tag_base;
                tag_base += nHS;
            }
            // few hits (late levels: a handful of buckets hold everything that is left): even small buckets get a wavefront
            // and LDS; many hits (tens of thousands of tiny buckets): those stay with one thread each
            g.lds_min = (u32)((nHS + nHP <= 32768 && !getenv("KC_LDS_MIN")) ? 4 : lds_min);
            g.abort2 = (bkS || bkP) ? w.d_flag : nullptr;
            // many events (the mid levels of a read set): the log goes through 64 cursors and is gathered afterwards
            const bool shlog = log_sharded && V >= ((u64)1 << 16) && !getenv("KC_NO_SHARDED_LOG") && local_any && nHS >= 4096;   // (small inputs: one bucket can outgrow a shard)
            g.ne_sh = shlog ? ne_sh : nullptr;
            if (shlog) KC_HIP(ctx, hipMemsetAsync(ne_shcur, 0, 65 * 16 * 8, ctx->stream));
            // after a read-back that carried the overflow words (hp[16], hp[17]): sort the overflowed side(s) again, LSD this time
            auto resort_overflowed = [&]() -> int {
                bool dummy;
                if (hp[16]) KC_TRY(sort_rows(w, hs_keys, hs_ids, nHS, W, d, hs_keys2, hs_ids2, sc0, sc1, sc2, sc3, 0, false, &dummy));
                if (hp[17]) KC_TRY(sort_rows(w, hp_keys, hp_ids, nHP, W, d, hp_keys2, hp_ids2, sc0, sc1, sc2, sc3, 1, false, &dummy));
                KC_HIP(ctx, hipMemsetAsync(w.d_flag, 0, 16, ctx->stream));
                bkS = bkP = false;
                return KC_OK;
            };
            if (local_any) {
                KC_LAUNCH((ov_match_kernel<W>), dim3(kc_grid(nHS, 64)), dim3(64), 0, ctx->stream, g, (const u64 *)hs_keys2,
                                   (const u64 *)hs_ids2, nHS, (const u64 *)hp_keys2, (const u64 *)hp_ids2, nHP, hs_ids, hp_ids, big, big_cap, cnt + 8, huge, huge_cap, cnt + 9);
                u64 nblk = (nHS + nHP) / (u64)g.lds_min + 1;
                if (nblk > big_cap) nblk = big_cap;
                KC_LAUNCH(ov_match_lds_kernel, dim3((unsigned)nblk), dim3(64), kc_lds_bytes_bound(KC_LDS_MID), ctx->stream, g,
                                   (const u64 *)hs_ids2, (const u64 *)hp_ids2, (const u64 *)big, big_cap, (const u64 *)(cnt + 8), 0u, (u32)KC_LDS_MID);
                if (nHS + nHP > KC_LDS_MID) {
                    KC_LAUNCH(ov_match_lds_kernel, dim3((unsigned)nblk), dim3(64), kc_lds_bytes_bound(KC_LDS_MAX), ctx->stream, g,
                                       (const u64 *)hs_ids2, (const u64 *)hp_ids2, (const u64 *)big, big_cap, (const u64 *)(cnt + 8), (u32)KC_LDS_MID, (u32)KC_LDS_MAX);
                }
                if (nHS + nHP > KC_LDS_MAX) {   // only then can a bucket be that large
                    u64 nh = (nHS + nHP) / KC_LDS_MAX + 1;
                    if (nh > huge_cap) nh = huge_cap;
                    KC_LAUNCH(ov_match_lds_kernel, dim3((unsigned)nh), dim3(64), kc_lds_bytes_bound(KC_LDS_HUGE), ctx->stream, g,
                                       (const u64 *)hs_ids2, (const u64 *)hp_ids2, (const u64 *)huge, huge_cap, (const u64 *)(cnt + 9), 0u, (u32)KC_LDS_HUGE);
                }
            }
            if (shlog) {
                const u64 per_shard_blocks = (2 * nHS / 64 * 5) / (256 * 8) + 1;
                KC_LAUNCH(ov_gather_log_kernel, dim3((unsigned)(per_shard_blocks > 1024 ? 1024 : per_shard_blocks), 64), dim3(256), 0, ctx->stream,
                                   (const u64 *)ne_sh, (const u64 *)ne_shcur, ne_shcap, ne, cnt);
            }
            const u64 *LOG = ne;   // the log the verification works on, and the counter block whose word 0 holds its length
            u64 *CNT = cnt;
            u64 covered = 0;
            if (!split) {
                if (!lower_bound)
                    KC_LAUNCH((ov_fix_kernel<false>), dim3(kc_grid(max_new, 256)), dim3(256), 0, ctx->stream, (const u64 *)ne, (const u64 *)cnt,
                                       (const u64 *)edge_from, (const u64 *)pred, first_snap, last_snap, cov, cnt + 2);
                KC_HIP(ctx, hipGetLastError());
                if (bkS || bkP) KC_HIP(ctx, hipMemcpyAsync(hp + 16, w.d_flag, 16, hipMemcpyDeviceToHost, ctx->stream));
                if (shlog) KC_HIP(ctx, hipMemcpyAsync(hp + 20, ne_shcur + 64 * 16, 8, hipMemcpyDeviceToHost, ctx->stream));
                KC_TRY(read_back(ctx, hp, cnt, 9));
                if (shlog && hp[20]) return kc_fail(ctx, KC_E_NO_DEVICE, "a shard of the new-edge log overflowed");
                if ((bkS || bkP) && (hp[16] | hp[17])) { KC_TRY(resort_overflowed()); continue; }   // (nothing was logged: the kernels stood down)
                nNew = hp[0];
                covered = hp[2];
            } else {
                // the ranks' logs are concatenated in rank order (the same everywhere); the others' edges are applied here
                u64 nloc = 0, counts[64], off_mine = 0;
                KC_HIP(ctx, hipGetLastError());
                if (local_any) {
                    if (bkS || bkP) KC_HIP(ctx, hipMemcpyAsync(hp + 16, w.d_flag, 16, hipMemcpyDeviceToHost, ctx->stream));
                    if (shlog) KC_HIP(ctx, hipMemcpyAsync(hp + 20, ne_shcur + 64 * 16, 8, hipMemcpyDeviceToHost, ctx->stream));
                    KC_TRY(read_back(ctx, hp, cnt, 9));
                    if (shlog && hp[20]) return kc_fail(ctx, KC_E_NO_DEVICE, "a shard of the new-edge log overflowed");
                    if ((bkS || bkP) && (hp[16] | hp[17])) { KC_TRY(resort_overflowed()); continue; }   // (local redo: no collective was entered yet)
                    nloc = hp[0];
                }
                KC_TRY(kcc_allgather_words(comm, &nloc, counts, 1));
                nNew = 0;
                for (u32 r = 0; r < PG; ++r) { if (r == pme) off_mine = nNew; nNew += counts[r]; }
                if (nNew) {
                    u64 so[64], sc[64], ro[64], rcv[64], a = 0;
                    for (u32 r = 0; r < PG; ++r) { so[r] = 0; sc[r] = nloc * 40; ro[r] = a * 40; rcv[r] = counts[r] * 40; a += counts[r]; }
                    KC_TRY(kcc_alltoallv(comm, ne, so, sc, ne_all, ro, rcv));
                    KC_LAUNCH(ov_apply_remote_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, (const u64 *)ne_all, nNew, off_mine,
                                       off_mine + nloc, d, edge_from, overlaps, pred);
                    CNT = fresh_cnt();
                    hp[32] = nNew;
                    KC_HIP(ctx, hipMemcpyAsync(CNT, hp + 32, 8, hipMemcpyHostToDevice, ctx->stream));
                    LOG = ne_all;
                    KC_LAUNCH((ov_fix_kernel<false>), dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, (const u64 *)CNT,
                                       (const u64 *)edge_from, (const u64 *)pred, first_snap, last_snap, cov, CNT + 2);
                    KC_HIP(ctx, hipGetLastError());
                    KC_TRY(read_back(ctx, hp + 40, CNT, 3));
                    covered = hp[42];
                }
            }
            if (nNew == 0 || lower_bound) break;
            if (covered == nNew) {
                KC_LAUNCH(ov_fix_write_clear_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, (const u64 *)CNT,
                                   (const u64 *)edge_from, (const u64 *)pred, first_snap, last_snap, cov);
                break;
            }
            // a cycle slipped through a stale path-end test: blacklist its earliest closing event, undo, replay
            ++repairs;
            if (h_bl.size() / 2 >= BL_MAX) return kc_fail(ctx, KC_E_NO_DEVICE, "overlap repair list exhausted");
            KC_LAUNCH(ov_index_edges_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, nNew, eidx);
            KC_HIP(ctx, hipMemsetAsync(CNT + 4, 0, 32, ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(CNT + 5, 0xff, 8, ctx->stream));
            KC_LAUNCH(ov_find_cycle_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, nNew,
                               (const u64 *)first_snap, (const u64 *)eidx, (const u8 *)cov, CNT + 4);
            KC_LAUNCH(ov_pick_event_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, nNew, CNT + 4);
            KC_TRY(read_back(ctx, hp, CNT + 4, 4));
            if (!hp[0]) return kc_fail(ctx, KC_E_NO_DEVICE, "overlap verification failed without a cycle");
            h_bl.push_back(hp[2]); h_bl.push_back(hp[3]);
            KC_HIP(ctx, hipMemcpyAsync(bl, h_bl.data(), h_bl.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            g.nbl = h_bl.size() / 2;
            KC_LAUNCH(ov_undo_kernel, dim3(kc_grid(nNew, 256)), dim3(256), 0, ctx->stream, LOG, nNew, edge_from, overlaps, pred,
                               eidx, cov);
        }
        if (trace) {
            (void)hipStreamSynchronize(ctx->stream); t_matched = now_us();
            fprintf(stderr, "[ov] d=%d nS=%llu hitS=%llu new=%llu big=%llu | probe %.0f  select+sort %.0f  match+fix %.0f us\n", d, (unsigned long long)nS,
                    (unsigned long long)nHS, (unsigned long long)nNew, (unsigned long long)hp[8], t_hits - t_level, t_sorted - t_hits, t_matched - t_sorted);
        }
        h_bl.clear(); g.nbl = 0;   // the blacklist is per level
        if (nNew == 0) return KC_OK;
        // 6. shrink the active lists -- lazily: every sweep checks for itself whether a listed vertex is still active, so the lists
        // are only compacted (two flag sweeps + two ordered compactions: 8 launches) once a quarter of their entries have died
        if (!lower_bound) { nS -= nNew; nP -= nNew; }   // every new edge removes exactly one suffix vertex and one prefix vertex
        if (!lower_bound && !getenv("KC_OV_EAGER_SHRINK") && (lenS - nS) * 4 <= lenS && (lenP - nP) * 4 <= lenP) return KC_OK;
        KC_LAUNCH((ov_still_active_kernel<true>), dim3(kc_grid(lenS, 256)), dim3(256), 0, ctx->stream, (const u64 *)actS, lenS,
                           (const u64 *)edge_from, (const u64 *)pred, hitS);
        KC_LAUNCH((ov_still_active_kernel<false>), dim3(kc_grid(lenP, 256)), dim3(256), 0, ctx->stream, (const u64 *)actP, lenP,
                           (const u64 *)edge_from, (const u64 *)pred, hitP);
        if (lower_bound) {   // an edge may have been logged twice: take the exact counts
            u64 nS2 = 0, nP2 = 0;
            KC_TRY(select_flagged(w, actS, hitS, actS2, lenS, &nS2));
            KC_TRY(select_flagged(w, actP, hitP, actP2, lenP, &nP2));
            nS = nS2; nP = nP2;
        } else {
            KC_TRY(select_flagged_nosync(w, actS, hitS, actS2, lenS));
            KC_TRY(select_flagged_nosync(w, actP, hitP, actP2, lenP));
        }
        lenS = nS; lenP = nP;
        std::swap(actS, actS2); std::swap(actP, actP2);
        return KC_OK;
    };
    // ---- the grouped levels (no host in the loop; see OVG_* above): every level that builds a table of its own, i.e. all but the
    // levels d >= 16 that go through the 16-mer table.  KC_OV_HOST_LOOP=1: the host-driven loop for everything.
    const bool grouped_ok = !lower_bound && !getenv("KC_OV_HOST_LOOP");
    u64 *ovg_st = nullptr, *ovg_lcnt = nullptr;
    OvgBuf ob{};
    bool lists_unsorted = false;
    if (grouped_ok) {
        KC_TRY(scope.alloc(&ovg_st, OVG_WORDS));
        KC_TRY(scope.alloc(&ovg_lcnt, 128 * 16));
        KC_TRY(scope.alloc(&ob.sslot, V + 1));
        KC_TRY(scope.alloc(&ob.boS, tcap + 1)); KC_TRY(scope.alloc(&ob.boP, tcap + 1));
        ob.cnt2 = reinterpret_cast<u32 *>(tab.slot + tcap);   // (the flag words of the host-driven levels)
        ob.pslot = pslot;
        ob.memS = hs_ids2; ob.memP = hp_ids2; ob.scrS = hs_ids; ob.scrP = hp_ids; ob.descA = hs_keys; ob.descB = hp_keys;
        ob.big = big; ob.huge = huge; ob.big_cap = big_cap; ob.huge_cap = huge_cap; ob.tcap = tcap;
        if (getenv("KC_OVG_NO_LDS")) { ob.big_cap = 0; ob.huge_cap = 0; }   // (test knob: every bucket a wavefront would replay sends its level to the host-driven loop)
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ovg_match_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kc_lds_bytes_bound(KC_LDS_HUGE)));
    }
    // levels d_hi .. 0 enqueued back to back, one look at the outcome; *next = the level the host-driven loop has to run (-1: none)
    auto run_grouped = [&](const int d_hi, int *next) -> int {
        *next = -1;
        u64 h_st[OVG_WORDS] = {0};
        h_st[OVG_LENS] = lenS; h_st[OVG_LENP] = lenP; h_st[OVG_LIVES] = nS; h_st[OVG_LIVEP] = nP;
        KC_HIP(ctx, hipMemcpyAsync(ovg_st, h_st, sizeof h_st, hipMemcpyHostToDevice, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (h_st is on the stack)
        ob.act[0] = actS; ob.act[1] = actS2; ob.act[2] = actP; ob.act[3] = actP2;
        // the new-edge log: through the 64 shards (gathered per level) on large inputs; straight into the dense log on small ones, where
        // one bucket can hold more events than a shard and an add per event on one word does not matter
        const bool shl = V >= ((u64)1 << 16) && ne_sh != nullptr;
        g.abort2 = nullptr; g.nbl = 0; g.lds_min = (u32)lds_min; g.ne_sh = shl ? ne_sh : nullptr;
        u64 *gcnt = nullptr;
        std::vector<u64 *> level_cnt((size_t)d_hi + 1, nullptr);
        KC_HIP(ctx, hipMemsetAsync(ovg_lcnt, 0, 128 * 16 * 8, ctx->stream));   // one counter block per level of this run (level d: block d)
        for (int d = d_hi; d >= 0; --d) {
            g.d = d;
            if (tag_base + nP + 2 >= 0xfffffff0ull) {   // (writer tags: a fresh range per level, never reused without clearing)
                KC_HIP(ctx, hipMemsetAsync(first, 0xff, V * 8, ctx->stream));
                KC_HIP(ctx, hipMemsetAsync(last, 0xff, V * 8, ctx->stream));
                tag_base = 0;
            }
            g.tag_base = (u32)tag_base;
            tag_base += nP + 1;   // (a level has at most as many buckets as live prefix vertices at the start of this run)
            gcnt = ovg_lcnt + 16 * (size_t)d;
            level_cnt[(size_t)d] = gcnt;
            g.ne_count = gcnt;
            KC_LAUNCH((ovg_clear_kernel<W>), dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, tab, ob.cnt2, ne_shcur, tcap);   // (+ the level's sizes, the previous level's swap)
            KC_LAUNCH((ovg_insert_kernel<W>), dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, tab, ob, (const u64 *)pre, (const u64 *)pred, k, d);
            KC_LAUNCH((ovg_probe_kernel<W>), dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, tab, ob, (const u64 *)suf, (const u64 *)pre, (const u64 *)edge_from, k, d);
            KC_LAUNCH((ovg_buckets_kernel<W>), dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, tab, ob, (const u64 *)pre, k, d, complements);
            KC_LAUNCH(ovg_scatter_kernel, dim3(OVG_GRID), dim3(256), 0, ctx->stream, (const u64 *)ovg_st, ob);
            KC_LAUNCH((ovg_match_kernel<W>), dim3(4 * OVG_GRID), dim3(64), 0, ctx->stream, g, ovg_st, ob, (const u64 *)suf);
            KC_LAUNCH(ovg_match_lds_kernel, dim3(8 * OVG_GRID), dim3(64), kc_lds_bytes_bound(KC_LDS_MID), ctx->stream, g, (const u64 *)ovg_st, ob, 0, 0u, (u32)KC_LDS_MID);
            KC_LAUNCH(ovg_match_lds_kernel, dim3(2 * OVG_GRID), dim3(64), kc_lds_bytes_bound(KC_LDS_MAX), ctx->stream, g, (const u64 *)ovg_st, ob, 0, (u32)KC_LDS_MID, (u32)KC_LDS_MAX);
            KC_LAUNCH(ovg_match_lds_kernel, dim3(64), dim3(64), kc_lds_bytes_bound(KC_LDS_HUGE), ctx->stream, g, (const u64 *)ovg_st, ob, 1, 0u, (u32)KC_LDS_HUGE);
            if (shl) KC_LAUNCH(ov_gather_log_kernel, dim3(16, 64), dim3(256), 0, ctx->stream, (const u64 *)ne_sh, (const u64 *)ne_shcur, ne_shcap, ne, gcnt);
            KC_LAUNCH(ovg_cover_kernel, dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, (const u64 *)ne, (const u64 *)gcnt, (const u64 *)edge_from, (const u64 *)pred,
                               (const u64 *)first_snap, (const u64 *)last_snap, cov);
            KC_LAUNCH(ovg_write_kernel, dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, (const u64 *)ne, (const u64 *)gcnt, (const u64 *)(ne_shcur + 64 * 16),
                               (const u64 *)edge_from, (const u64 *)pred, first_snap, last_snap, cov, d);   // (+ the level's verdict and accounts)
            KC_LAUNCH(ovg_compact_kernel, dim3(OVG_GRID), dim3(256), 0, ctx->stream, ovg_st, ob, (const u64 *)edge_from, (const u64 *)pred);
        }
        KC_LAUNCH(ovg_finish_kernel, dim3(1), dim3(1), 0, ctx->stream, ovg_st);
        KC_HIP(ctx, hipGetLastError());
        KC_HIP(ctx, hipMemcpyAsync(h_st, ovg_st, sizeof h_st, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_st[OVG_CUR] & 1) { std::swap(actS, actS2); std::swap(actP, actP2); }
        lenS = h_st[OVG_LENS]; lenP = h_st[OVG_LENP]; nS = h_st[OVG_LIVES]; nP = h_st[OVG_LIVEP];
        if (h_st[OVG_UNSORTED]) lists_unsorted = true;
        if (getenv("KC_OV_TRACE"))
            fprintf(stderr, "[ov] grouped levels %d..: status %llx, %llu edges, %llu / %llu live suffix / prefix vertices left\n", d_hi, (unsigned long long)h_st[OVG_STATUS],
                    (unsigned long long)h_st[OVG_EDGES], (unsigned long long)nS, (unsigned long long)nP);
        if (!h_st[OVG_STATUS]) return KC_OK;
        const int dp = (int)(h_st[OVG_STATUS] & 0xff) - 1, reason = (int)(h_st[OVG_STATUS] >> 8);
        if (reason == OVG_REASON_LOG) return kc_fail(ctx, KC_E_NO_DEVICE, "a shard of the new-edge log overflowed");
        // whatever level dp logged is taken back (a repair: everything; a bucket this path does not take: what the others matched)
        u64 *c2 = level_cnt[(size_t)dp];   // (the dense log's length, when the level logged straight into it)
        if (shl) {
            c2 = fresh_cnt();
            KC_LAUNCH(ov_gather_log_kernel, dim3(16, 64), dim3(256), 0, ctx->stream, (const u64 *)ne_sh, (const u64 *)ne_shcur, ne_shcap, ne, c2);
        }
        u64 nlog = 0;
        KC_HIP(ctx, hipMemcpyAsync(&nlog, c2, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (nlog) KC_LAUNCH(ov_undo_kernel, dim3(kc_grid(nlog, 256)), dim3(256), 0, ctx->stream, (const u64 *)ne, nlog, edge_from, overlaps, pred, eidx, cov);
        KC_HIP(ctx, hipGetLastError());
        *next = dp;
        return KC_OK;
    };
    // the host-driven levels want their lists ascending (the grouped ones compact them in any order)
    auto sort_lists = [&]() -> int {
        if (!lists_unsorted) return KC_OK;
        int vbits = 4;   // (vertex ids are below V: no more radix passes than their bits need -- 5 instead of 10 at 2 10^5 vertices)
        while (vbits < 40 && (V >> vbits) != 0) vbits += 4;
        KC_TRY(kcu_sort_pairs(ctx, actS, actS2, nullptr, nullptr, lenS, 0, vbits, false));
        KC_TRY(kcu_sort_pairs(ctx, actP, actP2, nullptr, nullptr, lenP, 0, vbits, false));
        std::swap(actS, actS2); std::swap(actP, actP2);
        lists_unsorted = false;
        return KC_OK;
    };
    // Which loop a level takes: the grouped levels win where a level is bound by launches and read-backs (few live vertices: 10^5
    // reads, the late levels of any input); with millions of live vertices a level is bound by its sweeps, and the host-driven one
    // makes fewer of them (measured at 5 10^6 vertices: 26.2 ms for the whole phase, 32.4 ms with every table level grouped, and
    // 26.8 - 27.4 ms with only the late levels grouped: tools/r4_ovgmax.sh).  KC_OVG_MAX = live vertices (suffix + prefix) up to
    // which the grouped levels are used.
    u64 ovg_max = (u64)1 << 20;
    if (const char *e = getenv("KC_OVG_MAX")) ovg_max = (u64)atoll(e);
    auto set_split = [&](bool on) { split = on; PG = on ? (u32)comm->world : 1u; pme = on ? (u32)comm->rank : 0u; };
    for (int d = k - 1; d >= 0 && nS > 0 && nP > 0;) {
        const bool t16_level = t16_on && !t16_dead && d >= 16;
        if (grouped_ok && !t16_level && nS + nP <= ovg_max) {
            int next = -1;
            set_split(false);
            KC_TRY(run_grouped(d, &next));
            if (next < 0) break;
            KC_TRY(sort_lists());
            KC_TRY(run_level_host(next));
            d = next - 1;
            continue;
        }
        // (t16_dead can only change inside a level that is not divided, so every rank sees the same t16_level here)
        set_split(can_split && !t16_level && nS + nP > ovg_max);
        KC_TRY(run_level_host(d));
        --d;
    }
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ok = true;
    out->V = V; out->edge_from = edge_from; out->overlaps = overlaps; out->pred = pred; out->repairs = repairs;
    scope.release(big); scope.release(huge);
    void *rel[] = {pre, suf, first, last, first_snap, last_snap, actS, actP, actS2, actP2, eidx, hitS, hitP, cov, pslot, ne, cnt_pool, bl, w.d_num,
                   tab.slot, hs_ids, hp_ids, hs_ids2, hp_ids2, hs_keys, hp_keys, hs_keys2, hp_keys2, sc0, sc1, sc2, sc3, w.tmp};
    for (void *p : rel) scope.release(p);
    return KC_OK;
}

int kco_overlap_path_device(kc_simplitigs *s, int complements, int lower_bound, KcPathDev *out, kc_comm *comm) {
    switch (s->W) {
        case 1: return overlap_path_w<1>(s, complements, lower_bound, out, comm);
        case 2: return overlap_path_w<2>(s, complements, lower_bound, out, comm);
        default: return overlap_path_w<4>(s, complements, lower_bound, out, comm);
    }
}
void kco_path_release(kc_ctx *ctx, KcPathDev *p) {
    kc_release(ctx, p->edge_from); kc_release(ctx, p->overlaps); kc_release(ctx, p->pred);
    p->edge_from = nullptr; p->overlaps = nullptr; p->pred = nullptr;
}

extern "C" int kc_overlap_path(kc_simplitigs *s, int complements, int lower_bound, uint64_t *edge_from, uint8_t *overlaps) {
    if (!s || !edge_from || !overlaps) return KC_E_ARG;
    kc_ctx *ctx = s->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    KcPathDev p;
    KC_TRY(kco_overlap_path_device(s, complements ? 1 : 0, lower_bound ? 1 : 0, &p));
    KC_HIP(ctx, hipMemcpyAsync(edge_from, p.edge_from, p.V * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipMemcpyAsync(overlaps, p.overlaps, p.V, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kco_path_release(ctx, &p);
    return KC_OK;
}

extern "C" int kc_lower_bound(kc_simplitigs *s, int complements, uint64_t *length) {   // lower_bound.h:7-21
    if (!s || !length) return KC_E_ARG;
    kc_ctx *ctx = s->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    KcPathDev p;
    KC_TRY(kco_overlap_path_device(s, complements ? 1 : 0, 1, &p));
    std::vector<u8> ov(p.V);
    KC_HIP(ctx, hipMemcpyAsync(ov.data(), p.overlaps, p.V, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kco_path_release(ctx, &p);
    // res = sum(simplitig.size()/(2-complements)) [size in bits] - sum(overlaps incl. 0xFF wrap like the reference's size_t arithmetic]
    u64 res = s->total_bases * 2 / (complements ? 1 : 2);
    for (u8 o : ov) res -= o;
    *length = res / (complements ? 2 : 1);
    return KC_OK;
}
