// Device code of phase 1 (greedy simplitigs by de Bruijn neighbour lookup) shared by the single-GPU flow
// (kc_simplitigs.hip) and the hash-sharded flow (kc_shard.hip): port-link words, the edge order of the canonical rule, the
// batched neighbour lookup of round 1.  See kc_simplitigs.hip for the algorithm.
#pragma once
#include "kc_internal.h"
#include "kc_scan_dev.h"
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <vector>

#define KC_MSPEC 16   /* minimizer length the k = 31 / 63 / 127 specialisations are compiled for (= kc_default_mlen) */
#define LK_OPEN 0ull
#define LK_CHOICE 1ull
#define LK_MATCHED 2ull
#define LK_DEAD 3ull
#define LK_CUT 4ull                   /* on a dead port: it was made by opening a cycle (not one of the ports the matching left dead) */
#define LK_STATE(w) ((w) & 3ull)
#define LK_PORT(w) (((w) >> 2) & 1ull)
#define LK_BASE(w) (((w) >> 3) & 3ull)
#define LK_REJ 32ull                  /* bit 5: the choice is not (yet) known to be reciprocated */
#define LK_SPLIT (1ull << 63)          /* the node of this port is a splitter: a segment start for the parallel chain walks */
#define LK_SLOT(w) ((((w) & ~LK_SPLIT) >> 6) & (((u64)1 << 36) - 1))   /* the target node (a dead port: the slot of its own k-mer); 36 bits */
/* bits 42-56 (hash-sharded flow): the offset of the word's OWN k-mer inside its table block -- with the block (node id / nstride) that
   is its slot, so whoever holds a port word can read the k-mer's key (the first k-mer of a chain fragment, whose entry port is
   dead or matched to a node of another rank) */
#define LK_OWN_SHIFT 42
#define LK_OWN(w) (((w) >> LK_OWN_SHIFT) & 0x7fffull)
#define LK_OWN_VALID (1ull << 57)      /* the word carries the offset (a word put back from the contraction log does not) */
#define LK_OWN_MASK ((0x7fffull << LK_OWN_SHIFT) | LK_OWN_VALID)
#define LK_OWN_BITS(off) (((u64)(off) << LK_OWN_SHIFT) | LK_OWN_VALID)
#define LK_PID(w) ((LK_SLOT(w) << 1) | LK_PORT(w))
#define LK_MAKE(state, port, base, slot) ((u64)(state) | ((u64)(port) << 2) | ((u64)(base) << 3) | ((u64)(slot) << 6))
// Round 1 needs no accept pass: the edge relation is symmetric, so a port's choice can only be unreciprocated if its
// target had at least two candidates; such "branch" ports (rare) are listed by the choose kernel and send explicit
// rejections (LK_REJ) to the candidates they did not choose.  A round-1 choice without LK_REJ is therefore mutual.
// Later rounds write their tentative choices WITH LK_REJ and confirm them to LK_MATCHED in kc_accept_kernel.
// Entries of the branch list and of the later rounds' work lists: the port id (dense node id << 1 | port) and, when the lister had it
// in hand, the slot offset of the port's k-mer inside its block -- the kernels that need the KEY of a listed port then skip the
// search for its slot (set_slot_of_nid: a dozen dependent reads).
#define KC_WK_HAS_OFF (1ull << 63)
#define KC_WK_PID(e) ((e) & (((u64)1 << 40) - 1))
#define KC_WK_TAG(off) ((((u64)(off)) << 40) | KC_WK_HAS_OFF)
#define LK_IS_MATCHED(w) (LK_STATE(w) == LK_MATCHED || (LK_STATE(w) == LK_CHOICE && !((w) & LK_REJ)))

// "edge (x,px)-(y1,q1) is preferred over edge (x,px)-(y2,q2)": higher priority first, ties by the ordered endpoint tuple.
template <int W>
__device__ __forceinline__ bool edge_preferred(const Key<W> &x, int px, const Key<W> &y1, int q1, u64 pr1, const Key<W> &y2, int q2,
                                               u64 pr2) {
    if (pr1 != pr2) return pr1 > pr2;
    // tuple (ka,pa,kb,pb) with (ka,pa) <= (kb,pb)
    bool x_first1 = key_lt(x, y1) || (key_eq(x, y1) && px <= q1);
    bool x_first2 = key_lt(x, y2) || (key_eq(x, y2) && px <= q2);
    const Key<W> &a1 = x_first1 ? x : y1; int pa1 = x_first1 ? px : q1;
    const Key<W> &b1 = x_first1 ? y1 : x; int pb1 = x_first1 ? q1 : px;
    const Key<W> &a2 = x_first2 ? x : y2; int pa2 = x_first2 ? px : q2;
    const Key<W> &b2 = x_first2 ? y2 : x; int pb2 = x_first2 ? q2 : px;
    if (!key_eq(a1, a2)) return key_lt(a1, a2);
    if (pa1 != pa2) return pa1 < pa2;
    if (!key_eq(b1, b2)) return key_lt(b1, b2);
    return pb1 < pb2;
}

// A port that ends DEAD (no partner left) is the entry port of a directed chain: oriented node (slot, o = port).  Every
// port ends either MATCHED or DEAD, so the heads are exactly the ports that die (unidirectional mode: the L ports).
__device__ __forceinline__ void kc_push_head(u64 pid, int complements, u64 *heads, u64 heads_cap, u64 *heads_cursor) {
    if (!complements && (pid & 1)) return;
    u64 pos = atomicAdd((ull *)heads_cursor, 1ULL);
    if (pos < heads_cap) heads[pos] = pid;
}

// Round 1 lists its heads and branch ports through KC_NSHARD cursors (one 128-byte line each, picked by workgroup): chain
// heads are sparse (2 per simplitig), so the compiler's wave-level aggregation of the atomic leaves ~1 atomic per event,
// and ~10^5..10^6 atomics on ONE address serialise at ~13 ns each -- longer than the kernel's real work.  The sharded
// lists are gathered into one contiguous list afterwards (kc_gather_shards_kernel).
#define KC_NSHARD 64
#define KC_SHARD_STRIDE 16
__device__ __forceinline__ void kc_push_sharded(u64 val, u64 *list, u64 shard_cap, u64 *cursors) {
    const u64 sh = blockIdx.x & (KC_NSHARD - 1);
    u64 pos = atomicAdd((ull *)&cursors[sh * KC_SHARD_STRIDE], 1ULL);
    if (pos < shard_cap) list[sh * shard_cap + pos] = val;
}
// grid (x, KC_NSHARD): shard blockIdx.y is copied behind the shards before it; *total_out = total, or dst_cap + 1 when a
// shard (or the destination) overflowed, which the caller treats like any other list overflow.
static __global__ void __launch_bounds__(256) kc_gather_shards_kernel(const u64 *__restrict__ src, u64 shard_cap, const u64 *__restrict__ cursors,
                                                               u64 *dst, u64 dst_cap, u64 *total_out) {
    __shared__ u64 pre[KC_NSHARD + 1];
    __shared__ int ovf;
    if (threadIdx.x == 0) {
        u64 a = 0;
        int o = 0;
        for (int sh = 0; sh < KC_NSHARD; ++sh) {
            u64 c = cursors[sh * KC_SHARD_STRIDE];
            if (c > shard_cap) { o = 1; c = shard_cap; }
            pre[sh] = a;
            a += c;
        }
        pre[KC_NSHARD] = a;
        ovf = o || a > dst_cap;
    }
    __syncthreads();
    const u64 sh = blockIdx.y, base = pre[sh], n = pre[sh + 1] - pre[sh];
    if (!ovf)
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) dst[base + i] = src[sh * shard_cap + i];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *total_out = ovf ? dst_cap + 1 : pre[KC_NSHARD];
}

// The 8 candidate lookups of one k-mer on a table with PLAIN hashing (set.m == 0, i.e. k < 18: every candidate has a home of
// its own; tables with minimizer blocks go through kc_cand_lookup_mid below).
#define KC_REMOTE (~(u64)1)
template <int W> struct KcCand {
    Key<W> x;        // the k-mer itself
    Key<W> cy[8];    // candidate i = port (i >> 2), extension character (i & 3): canonical key
    u64 cslot[8];    // its slot, KC_NONE64 (absent or not a legal edge), or KC_REMOTE
    u64 chome[8];    // home slot of the candidate key
    u32 qbits;       // bit i: the port on which candidate i is entered
    bool xpal;       // x is its own reverse complement (links on port L only)
};
template <int W, int KC, bool SH>
__device__ __forceinline__ void kc_cand_lookup(const SetView<W> &set, int k_rt, int complements, const u64 slot, const u64 *stage, const u64 sblk,
                                               KcCand<W> &out) {
    const int k = KC ? KC : k_rt;
    const int m = KC ? KC_MSPEC : set.m;
    const Key<W> x = set_key_at(set, slot);
    const Key<W> xr = key_rc(x, k);
    const bool xpal = (KC & 1) ? false : (complements && key_eq(x, xr));   // odd k: no palindromes
    u32 sharedR = ~0u, sharedL = ~0u;   // min over m-mer positions 1..k-m (append side) / 0..k-m-1 (prepend side)
    u32 f_first = 0, r_first = 0, f_last = 0, r_last = 0;   // first / last m-mer of x, forward and reverse complement
    if (m) {
        const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
        const int top = 2 * (m - 1);
        u32 f = 0, r = 0;
        if constexpr (W != 1) {
#pragma unroll
            for (int j = 0; j < m - 1; ++j) {
                u32 b = key_at(x, k, j);
                f = (f << 2) | b;
                r = (r >> 2) | ((3u - b) << top);
            }
        }
        u32 first = ~0u, mid = ~0u, last = ~0u;
#pragma unroll
        for (int i = 0; i <= k - m; ++i) {
            if constexpr (W == 1) {   // one word: every m-mer is a shift of x / RC(x), no serial dependency between positions
                f = (u32)(x.w[0] >> (2 * (k - m - i))) & maskm;
                r = (u32)(xr.w[0] >> (2 * i)) & maskm;
            } else {
                u32 b = key_at(x, k, i + m - 1);
                f = ((f << 2) | b) & maskm;
                r = (r >> 2) | ((3u - b) << top);
            }
            u32 h = kc_mmer_hash(f, r);
            if (i == 0) { first = h; f_first = f; r_first = r; }
            if (i == k - m) last = h;
            if (i != 0 && i != k - m) mid = h < mid ? h : mid;
        }
        f_last = f; r_last = r;
        if (k - m == 0) { sharedR = ~0u; sharedL = ~0u; }
        else { sharedR = mid < last ? mid : last; sharedL = mid < first ? mid : first; }
    }
    // The 8 candidate lookups are independent, so they are issued as a batch: (A) all candidate keys and their homes,
    // (B) all first tag windows (one unaligned 8-byte load each), (C) the key loads of the tag hits, (D) compares.  A
    // lookup that the first window does not settle (tag false positive, probe run longer than 8) is rare; those are
    // redone by the generic scan in ONE loop after the batch, so a wave pays that latency once, not once per candidate.
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const int top = 2 * (m - 1);
    // candidate (port, c) of the k-mer described by cx: canonical key y, the port q it is entered on, placement hash and home
    struct Ctx { Key<W> x, xr; u32 f_first, r_first, f_last, r_last, sharedR, sharedL; };
    auto make_cand_of = [&](const Ctx &cx, const int port, const u32 c, Key<W> &y, int &q, u64 &hy, u64 &home) {
        Key<W> nf = port ? key_append(cx.x, (u64)c, k) : key_prepend(cx.x, (u64)c, k);
        y = nf;
        if (complements) {
            Key<W> r = port ? key_prepend(cx.xr, (u64)(3 - c), k) : key_append(cx.xr, (u64)(3 - c), k);   // RC(nf)
            bool pal = (KC & 1) ? false : key_eq(nf, r);
            bool fwd = !key_lt(r, nf);
            if (!fwd) y = r;
            q = pal ? 0 : (port ? (fwd ? 0 : 1) : (fwd ? 1 : 0));
        } else {
            q = port ? 0 : 1;
        }
        hy = key_place_hash(y);
        if (m) {
            // the one m-mer the candidate does not share with x, one shift away from x's last / first m-mer
            u32 nfw = port ? (((cx.f_last << 2) | c) & maskm) : ((cx.f_first >> 2) | (c << top));
            u32 nrc = port ? ((cx.r_last >> 2) | ((3u - c) << top)) : (((cx.r_first << 2) | (3u - c)) & maskm);
            u32 h = kc_mmer_hash(nfw, nrc);
            const u32 shared = port ? cx.sharedR : cx.sharedL;
            home = set_block_home(set, h < shared ? h : shared, hy);
        } else home = kc_mulhi64(kc_mix64(hy), set.cap);
    };
    const Ctx own{x, xr, f_first, r_first, f_last, r_last, sharedR, sharedL};
    auto make_cand = [&](const int port, const u32 c, Key<W> &y, int &q, u64 &hy, u64 &home) { make_cand_of(own, port, c, y, q, hy, home); };
    Key<W> (&cy)[8] = out.cy;
    u64 (&chome)[8] = out.chome, (&cslot)[8] = out.cslot;
    u64 cph[8], cword[8];
    u32 live = 0, qbits = 0, remote = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int q;
        make_cand(i >> 2, (u32)(i & 3), cy[i], q, cph[i], chome[i]);
        const bool ok = !key_eq(cy[i], x) && !((i >> 2) == 1 && xpal);   // no self edges; palindromes link on port L only
        live |= (u32)ok << i;
        qbits |= (u32)q << i;
    }
    // first tag window of every candidate: from the staged block (LDS) when the home lies in it, else from global memory
    u32 inblk = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 in = (chome[i] >> set.bshift) == sblk ? 1u : 0u;
        inblk |= in << i;
        if (SH) {   // homes in another rank's range are not addressable here: read x's own tag window instead (harmless), note the candidate
            const u32 rem = (chome[i] - set.rlo >= set.rcap) ? 1u : 0u;
            remote |= (rem & (live >> i)) << i;
            if (rem) live &= ~(1u << i);
            if (!in) __builtin_memcpy(&cword[i], set.tags + (rem ? slot : chome[i]), 8);
        } else if (!in) __builtin_memcpy(&cword[i], set.tags + chome[i], 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((inblk >> i) & 1) {
            // one unaligned 8-byte LDS read (gfx950 reads LDS at any byte offset; the stage has 8 bytes of slack at its end)
            const u32 o = (u32)chome[i] & ((1u << set.bshift) - 1u);
            __builtin_memcpy(&cword[i], reinterpret_cast<const u8 *>(stage) + o, 8);
        }
    // branch-free on purpose: a conditional here lets the compiler sink a candidate's load into the branch and wait for it.
    // The 8 tag bytes of a window are searched in 32-bit halves: "byte equals the tag" and "byte is empty" each become one bit
    // per byte (bit 7), and a multiply gathers the four bits of a half into a nibble -- (h * 0x204081) >> 28 -- so that
    // everything after that (first candidate, is it a match, is the next one an empty slot) works on two 8-bit masks.
    u32 pending = 0, slow = 0, settled2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 lo7 = 0x7f7f7f7fu, hi7 = 0x80808080u, gather = 0x204081u;
        const u32 pat = (u32)kc_tag_of(cph[i]) * 0x01010101u;
        const u32 wl = (u32)cword[i], wh = (u32)(cword[i] >> 32);
        const u32 xl = wl ^ pat, xh = wh ^ pat;
        const u32 ml = ~(((xl & lo7) + lo7) | xl) & hi7, mh = ~(((xh & lo7) + lo7) | xh) & hi7;   // bit 7 of the bytes that equal the tag
        const u32 el = ~wl & hi7, eh = ~wh & hi7;                                                  // bit 7 of the empty bytes (tag 0)
        const u32 M8 = ((ml * gather) >> 28) | (((mh * gather) >> 28) << 4);
        const u32 E8 = ((el * gather) >> 28) | (((eh * gather) >> 28) << 4);
        u32 C = M8 | E8;
        if (set.local_probes < 8) C &= (1u << set.local_probes) - 1u;   // an empty slot proves a miss only inside the local probe run
        const u32 lv = (live >> i) & 1u;
        const u32 none = C == 0 ? 1u : 0u;                          // window full of other tags: keep scanning (pooled redo)
        const u32 b = (u32)__builtin_ctz(C | 0x100u);
        const u32 hit = (M8 >> b) & 1u;                             // first candidate is a tag match, not an empty slot (0 when none)
        u64 sl = chome[i] + (u64)b;
        if (!SH) sl = sl >= set.cap ? sl - set.cap : sl;   // (plain hashing only; block homes keep their window inside the block)
        const u32 pend = lv & hit;
        // if that tag match turns out to be another key, the next candidate of the window usually is the empty slot
        // that proves the miss: note it now, so that a tag false positive does not cost a trip through the redo
        const u32 C2 = C & (C - 1u);
        const u32 b2 = (u32)__builtin_ctz(C2 | 0x100u);
        const u32 miss2 = (E8 >> b2) & 1u;                          // (0 when there is no second candidate)
        slow |= (lv & none) << i;
        pending |= pend << i;
        settled2 |= miss2 << i;
        cslot[i] = pend ? sl : slot;   // non-pending candidates re-read x's own slot: an unconditional, harmless load
    }
    Key<W> ck[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ck[i] = set_key_at(set, cslot[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 pend = (pending >> i) & 1u;
        const u32 eq = key_eq(ck[i], cy[i]) ? 1u : 0u;
        slow |= (pend & (eq ^ 1u) & (((settled2 >> i) & 1u) ^ 1u)) << i;   // tag false positive, miss not proven yet: slow path
        cslot[i] = (pend & eq) ? cslot[i] : KC_NONE64;
    }
    // Lookups the first window did not settle (3.5 % of them, i.e. ~18 per wavefront and sweep step: nearly every wave has
    // some, few lanes have more than one).  A per-lane redo loop made the whole wave pay for its unluckiest lane, once per
    // trip; instead the wave pools them: every lane lists its unsettled (lane, candidate) pairs in LDS, lane e redoes entry e
    // -- it pulls the owner's k-mer context with shuffles, rebuilds the candidate and runs the generic scan from the known
    // home -- and the owners read the slots back.  One dense trip per 64 unsettled lookups.
    // (The lanes that are active here are a prefix of the wave: the sweep loops run over t = t0 + threadIdx.x < n.)
    if (__ballot(slow != 0)) {
        __shared__ unsigned short kc_sq_ent[4][512];
        __shared__ u64 kc_sq_res[4][512];
        const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
        const u32 nact = (u32)__popcll(__ballot(1));
        const u32 cnt = (u32)__popc(slow);
        const unsigned long long lt = ((unsigned long long)1 << lane) - 1;
        u32 pre = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // exclusive prefix sum of cnt (0..8) over the active lanes, by bit planes
            const unsigned long long mj = __ballot((cnt >> j) & 1u);
            pre += (u32)__popcll(mj & lt) << j;
            total += (u32)__popcll(mj) << j;
        }
        {
            u32 sl = slow, at = pre;
            while (sl) {
                const int i = __builtin_ctz(sl);
                sl &= sl - 1;
                kc_sq_ent[wv][at++] = (unsigned short)((lane << 3) | i);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (u32 r0 = 0; r0 < total; r0 += nact) {
            const u32 e = r0 + (u32)lane;
            const bool has = e < total;
            const u32 ent = kc_sq_ent[wv][has ? e : 0];
            const int ol = (int)(ent >> 3);
            Ctx cx;   // the owner's context: every active lane takes part in the shuffles, the owners are among them
#pragma unroll
            for (int w = 0; w < W; ++w) { cx.x.w[w] = __shfl(x.w[w], ol, 64); cx.xr.w[w] = __shfl(xr.w[w], ol, 64); }
            cx.f_first = __shfl(f_first, ol, 64); cx.r_first = __shfl(r_first, ol, 64);
            cx.f_last = __shfl(f_last, ol, 64); cx.r_last = __shfl(r_last, ol, 64);
            cx.sharedR = __shfl(sharedR, ol, 64); cx.sharedL = __shfl(sharedL, ol, 64);
            if (has) {
                Key<W> y; int q; u64 hy, home;
                make_cand_of(cx, (int)((ent >> 2) & 1), ent & 3u, y, q, hy, home);
                kc_sq_res[wv][ent] = set_find_at(set, y, KcPlace{home, hy, kc_tag_of(hy)});   // (this lookup only serves tables with plain hashing)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((slow >> i) & 1u) cslot[i] = kc_sq_res[wv][(lane << 3) | i];
    }
    if (SH) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if ((remote >> i) & 1) cslot[i] = KC_REMOTE;
    }
    out.x = x; out.qbits = qbits; out.xpal = xpal;
}

// candidate (port, c) of x: canonical key and the port q it is entered on (c = the character appended on port R / prepended on port L)
template <int W> __device__ __forceinline__ Key<W> kc_neighbour_key_q(const Key<W> &x, const Key<W> &xr, int port, u32 c, int k, int complements, int &q) {
    const Key<W> nf = port ? key_append(x, (u64)c, k) : key_prepend(x, (u64)c, k);
    if (!complements) { q = port ? 0 : 1; return nf; }
    const Key<W> r = port ? key_prepend(xr, (u64)(3 - c), k) : key_append(xr, (u64)(3 - c), k);   // RC(nf)
    const bool pal = key_eq(nf, r), fwd = !key_lt(r, nf);
    q = pal ? 0 : (port ? (fwd ? 0 : 1) : (fwd ? 1 : 0));   // append: entering y on L if y == nf else R; prepend: the other way round; palindromes: L
    return fwd ? nf : r;
}
template <int W> __device__ __forceinline__ Key<W> kc_neighbour_key(const Key<W> &x, int port, u32 c, int k, int complements) {
    int q;
    return kc_neighbour_key_q(x, complements ? key_rc(x, k) : x, port, c, k, complements, q);
}

// ---- Neighbour lookup on a table with middle-based homes (set.m != 0; kc_set_dev.h).  The 4 candidates of a port share their
// middle (k-2)-mer, hence ONE home, one placement hash and one tag window; the tag bytes carry the outer characters, so the
// window search says which extensions can exist and only those keys are compared.  Per k-mer: one pass over the k-m+1 m-mer
// hashes (the minimizers of both candidate middles are a suffix / a prefix minimum of them), two placement hashes, two
// homes, two window searches -- against eight of each with per-key homes.
struct KcPortDesc {
    u64 home;        // the home the 4 candidates of the port share
    u32 pat, msk;    // tag pattern of the candidates and the bits of it that are fixed (the free field is the extension)
    u32 flags;       // bit 0: the middle is read reversed; bit 1: the fixed outer character is field a; bit 2: palindromic middle (generic lookups)
};
// descriptor of port `port` (0 = L / prepend side, 1 = R / append side) of x; mmh = minimizer hash of the candidates' middle
// LOCAL: the home is taken INSIDE the block that starts at slot `local_base` (offset = range reduction of the middle's hash, as
// set_block_home has it) and no minimizer is needed: the guess "the candidates live in my own block" of the GP probe kernel.
template <int W, int KC, bool LOCAL = false>
__device__ __forceinline__ KcPortDesc kc_port_desc(const SetView<W> &set, const Key<W> &x, const Key<W> &xr, int k, int complements, int port, u32 mmh,
                                                   const u64 local_base = 0) {
    const Key<W> lm = key_lowmask<W>(2 * (k - 2));
    const Key<W> mf = port ? key_and(x, lm) : key_shr(x, 4);
    const Key<W> mr = port ? key_shr(xr, 4) : key_and(xr, lm);
    const bool rev = complements && key_lt(mr, mf);
    const bool tie = !(KC & 1) && complements && key_eq(mr, mf);   // palindromic middle (even k only): the tags say nothing
    const Key<W> M = rev ? mr : mf;
    const u64 mh = key_place_hash(M);
    // the character the candidates share next to the middle: x[1] (append side) / x[k-2] (prepend side); reading the candidate in
    // the middle's canonical orientation it is the FIRST outer character (field a) for append & forward or prepend & reversed,
    // else the last one (field b); a reversed reading complements it.  The other field is the extension.
    const u32 fx = port ? key_at(x, k, 1) : (u32)((x.w[0] >> 2) & 3);
    const bool fa = port ? !rev : rev;
    const u32 fval = rev ? 3u - fx : fx;
    KcPortDesc d;
    d.home = LOCAL ? (local_base | (((u64)(u32)mh * set_block_span(set)) >> 32)) : set_block_home(set, mmh, mh);
    d.pat = 0x80u | ((u32)mh & 7u) | (fa ? fval << 5 : fval << 3);
    d.msk = fa ? 0xE7u : 0x9Fu;
    d.flags = (rev ? 1u : 0u) | (fa ? 2u : 0u) | (tie ? 4u : 0u);
    return d;
}
// minimizer hash of a middle given forward (mid) and reverse complement (midr), both (k-2)-mers
template <int W, int KC>
__device__ __forceinline__ u32 kc_mid_min(const Key<W> &mid, const Key<W> &midr, int k, int m) {
    const int k2 = k - 2;
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    u32 best = ~0u;
    if constexpr (W == 1) {
#pragma unroll
        for (int i = 0; i <= k2 - m; ++i) {
            const u32 f = (u32)(mid.w[0] >> (2 * (k2 - m - i))) & maskm, r = (u32)(midr.w[0] >> (2 * i)) & maskm;
            const u32 h = kc_mmer_hash(f, r);
            best = h < best ? h : best;
        }
    } else {
        best = key_min_mmh(mid, k2, m, 0, k2 - m);
        (void)midr;
    }
    return best;
}
// generic lookups of the 4 candidates of a port (palindromic middles, overfull blocks whose keys went to their second home)
template <int W>
__device__ __forceinline__ void kc_port_search_generic(const SetView<W> &set, int k, int complements, const Key<W> &x, const Key<W> &xr, int port,
                                                       u64 (&cs)[4], u32 &qb) {
    qb = 0;
#pragma unroll
    for (u32 c = 0; c < 4; ++c) {
        int q;
        const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
        cs[c] = KC_NONE64;
        if (key_eq(y, x)) continue;
        const u64 sl = set_find(set, y);
        cs[c] = sl;
        if (sl != KC_NONE64) qb |= (u32)q << c;
    }
}
// Search the probe run of one port.  cs[c] = slot of the candidate with extension c (appended on port R, prepended on port L) or
// KC_NONE64; bit c of qb = the port that candidate is entered on.  Returns false when the window search cannot settle the port
// (palindromic middle, local probe run exhausted): the caller then uses kc_port_search_generic, or hands the port on.
//   LT: the home lies in block `sblk` whose tags are staged in LDS (stage_tags); SK: that block's keys are staged too.
//   LT == 2: decided per port -- a home in block `sblk` is read from the staged copies (through flat pointers), any other from the table
//   DN: *eoff = offset from the home of the empty slot E that ended the run.  The k-mers between a match and E are all present, so
//   a candidate's dense node id is block * nstride + count(E) - (E - slot); count(E) sits in the key word of E, next to the keys the
//   search has just read (kc_port_delta)
template <int W, int LT, bool SK, bool DN = false>
__device__ __forceinline__ bool kc_port_search(const SetView<W> &set, int k, int complements, const Key<W> &x, const Key<W> &xr, const u64 slot, int port,
                                               const KcPortDesc &d, const u8 *stage_tags, const u64 *stage_keys, u64 (&cs)[4], u32 &qb, const u64 sblk = ~(u64)0,
                                               u32 *eoff = nullptr) {
    const u32 bmask = (1u << set.bshift) - 1u;
#pragma unroll
    for (int c = 0; c < 4; ++c) cs[c] = KC_NONE64;
    qb = 0;
    bool slow = (d.flags & 4u) != 0;
    if (!slow) {
        const u32 lo7 = 0x7f7f7f7fu, hi7 = 0x80808080u, gather = 0x204081u;
        const u32 pat4 = d.pat * 0x01010101u, msk4 = d.msk * 0x01010101u;
        const bool staged = LT == 1 || (LT == 2 && (d.home >> set.bshift) == sblk);
        const u8 *tg = staged ? stage_tags + ((u32)d.home & bmask) : set.tags + d.home;
        const u64 *kb = (SK && staged) ? stage_keys : set.keys;
        u32 off = 0;
        while (true) {
            u64 wd;
            __builtin_memcpy(&wd, tg + off, 8);   // 8 tags in one unaligned load (gfx950 reads LDS at any byte offset too)
            // "byte matches the pattern on the fixed bits" and "byte is empty" become one bit per byte (bit 7), and a multiply gathers
            // the four bits of a half into a nibble -- (h * 0x204081) >> 28 -- so that everything else works on two 8-bit masks
            const u32 wl = (u32)wd, wh = (u32)(wd >> 32);
            const u32 xl = (wl ^ pat4) & msk4, xh = (wh ^ pat4) & msk4;
            const u32 ml = ~(((xl & lo7) + lo7) | xl) & hi7, mh4 = ~(((xh & lo7) + lo7) | xh) & hi7;
            const u32 el = ~wl & hi7, eh = ~wh & hi7;
            const u32 M8 = ((ml * gather) >> 28) | (((mh4 * gather) >> 28) << 4);
            u32 E8 = ((el * gather) >> 28) | (((eh * gather) >> 28) << 4);
            const u32 left = set.local_probes - off;                 // slots of the local probe run still ahead
            if (left < 8) E8 |= 1u << left;                           // the run ends inside this window: treated like an empty slot, see below
            const u32 fe = (u32)__builtin_ctz(E8 | 0x100u);          // first empty slot of the window (8: none)
            u32 H = M8 & ((1u << fe) - 1u);                           // no deletions: a key lies before the first empty slot after its home
            while (H) {
                const u32 j = (u32)__builtin_ctz(H);
                H &= H - 1u;
                const u64 sl = d.home + (u64)(off + j);
                if (sl == slot) continue;                             // x itself (no self edges)
                Key<W> kk;
                if (LT == 1 && SK) {
#pragma unroll
                    for (int w = 0; w < W; ++w) kk.w[w] = stage_keys[(size_t)((u32)sl & bmask) * W + w];
                } else if (LT == 2 && SK) {
                    const size_t ki = (size_t)(staged ? (u64)((u32)sl & bmask) : sl) * W;
#pragma unroll
                    for (int w = 0; w < W; ++w) kk.w[w] = kb[ki + w];
                } else kk = set_key_at(set, sl);
                u32 c, q;
                if (kc_neighbour_match(kk, x, xr, port, k, complements, c, q)) {
#pragma unroll
                    for (u32 cc = 0; cc < 4; ++cc)
                        if (c == cc) cs[cc] = sl;                     // (selects: a run-time index would send the array to scratch memory)
                    qb |= q << c;
                }
            }
            if (fe < 8 && !(left < 8 && fe == left)) {               // a real empty slot: every candidate of this home has been seen
                if (DN) *eoff = off + fe;   // (the caller turns it into the id - slot difference: kc_port_delta)
                break;
            }
            if (left <= 8) { slow = true; break; }                    // local run exhausted: the rest may sit at the second home
            off += 8;
        }
    }
    return !slow;
}

// (dense node id - slot) of the k-mers that sit between the home of a port and the empty slot at home + eoff that ended its run
template <int W, bool SK>
__device__ __forceinline__ u64 kc_port_delta(const SetView<W> &set, const u64 home, const u32 eoff, const bool staged, const u64 *stage_keys) {
    const u32 bmask = (1u << set.bshift) - 1u;
    const u64 eabs = home + (u64)eoff;
    const u64 ann = (SK && staged) ? stage_keys[(size_t)((u32)eabs & bmask) * W] : set.keys[eabs * W];
    return (home >> set.bshift) * (u64)set.nstride + KC_ANN_COUNT(ann) - eabs;
}
// Is the stored key kk a neighbour of x on `port`?  A neighbour is x shifted by one character with the extension c at the free
// end, or the reverse complement of that: kk XOR (the shifted x, or the shifted RC(x)) is then zero outside the extension
// position.  (Stored keys are canonical, so a match with either strand is the canonical candidate; no candidate key is built.)
// Returns found, the extension character c (appended on port R / prepended on port L) and the port q that kk is entered on.
template <int W>
__device__ __forceinline__ bool kc_neighbour_match(const Key<W> &kk, const Key<W> &x, const Key<W> &xr, int port, int k, int complements, u32 &c, u32 &q) {
    const Key<W> lowm = key_lowmask<W>(2 * k);
    // the strand whose extension sits at the LOW end: x (append side) / RC(x) (prepend side); the other strand has it at the top
    const Key<W> &sl = port ? x : xr, &sh = port ? xr : x;
    Key<W> dl = key_and(key_shl(sl, 2), lowm), dh = key_shr(sh, 2);
#pragma unroll
    for (int w = 0; w < W; ++w) { dl.w[w] ^= kk.w[w]; dh.w[w] ^= kk.w[w]; }
    const u32 cl = (u32)dl.w[0] & 3u;
    const int ts = 2 * (k - 1), tw = ts >> 6;
    u64 hv = dh.w[0];
#pragma unroll
    for (int w = 1; w < W; ++w) hv = (tw == w) ? dh.w[w] : hv;
    const u32 ch = (u32)(hv >> (ts & 63)) & 3u;
    dl.w[0] &= ~(u64)3;
    bool lz = true, hz = true;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        lz = lz && dl.w[w] == 0;
        hz = hz && (dh.w[w] & ~((tw == w) ? ((u64)3 << (ts & 63)) : (u64)0)) == 0;
    }
    // append side: forward = low match (c = cl), reverse = high match (c = 3 - ch); prepend side: forward = high match (c = ch),
    // reverse = low match (c = 3 - cl)
    const bool fwd = port ? lz : hz, rev = complements && (port ? hz : lz);
    const u32 cf = port ? cl : ch, cr = 3u - (port ? ch : cl);
    c = fwd ? cf : cr;
    q = (fwd && rev) ? 0u : (fwd ? (port ? 0u : 1u) : (port ? 1u : 0u));   // entering y on L if it is the forward strand of an append (...); palindromes: L
    return fwd || rev;
}

// The common case of kc_port_search for a home in the staged block, without loops: ONE window, settled iff it holds an empty slot
// with at most one tag match before it (then that key decides).  Anything else -- no empty slot among the 8, a second match (a tag
// false positive or a branch), a palindromic middle -- returns false and the port is looked up by the loop version later, pooled
// with the ports of other k-mers: inside the sweep a second loop trip for 2 of 64 lanes costs the whole wave a trip.
// DN: sl comes back as the dense node id of the match (the home lies in the staged block, whose first id is nb0): the slots from
// the match to the window's first empty slot are all occupied, and that empty slot's key word holds the count before it.
template <int W, bool SK, bool DN = false>
__device__ __forceinline__ bool kc_port_search_one(const SetView<W> &set, int k, int complements, const Key<W> &x, const Key<W> &xr, const u64 slot, int port,
                                                   const KcPortDesc &d, const u8 *stage_tags, const u64 *stage_keys, bool &found, u32 &c, u32 &q, u64 &sl, const u64 nb0 = 0) {
    const u32 bmask = (1u << set.bshift) - 1u;
    const u32 lo7 = 0x7f7f7f7fu, hi7 = 0x80808080u, gather = 0x204081u;
    const u32 pat4 = d.pat * 0x01010101u, msk4 = d.msk * 0x01010101u;
    u64 wd;
    __builtin_memcpy(&wd, stage_tags + ((u32)d.home & bmask), 8);
    const u32 wl = (u32)wd, wh = (u32)(wd >> 32);
    const u32 xl = (wl ^ pat4) & msk4, xh = (wh ^ pat4) & msk4;
    const u32 ml = ~(((xl & lo7) + lo7) | xl) & hi7, mh4 = ~(((xh & lo7) + lo7) | xh) & hi7;
    const u32 el = ~wl & hi7, eh = ~wh & hi7;
    const u32 M8 = ((ml * gather) >> 28) | (((mh4 * gather) >> 28) << 4);
    u32 E8 = ((el * gather) >> 28) | (((eh * gather) >> 28) << 4);
    if (set.local_probes < 8) E8 = 0;                               // (test knob: short local runs always take the loop version)
    const u32 fe = (u32)__builtin_ctz(E8 | 0x100u);                // first empty slot of the window (8: none)
    const u32 H = M8 & ((1u << fe) - 1u);
    found = false; c = 0; q = 0; sl = slot;
    if ((d.flags & 4u) || fe == 8 || (H & (H - 1u))) return false;
    const u32 j = (u32)__builtin_ctz(H | 0x100u);                  // the one match, if any (8: none -- then x's own key is read: harmless)
    sl = H ? d.home + (u64)j : slot;
    Key<W> kk;
    if (SK) {
#pragma unroll
        for (int w = 0; w < W; ++w) kk.w[w] = stage_keys[(size_t)((u32)sl & bmask) * W + w];
    } else kk = set_key_at(set, sl);
    found = kc_neighbour_match(kk, x, xr, port, k, complements, c, q) && H != 0 && sl != slot;   // (sl == slot: x itself, no self edges)
    if (DN) {
        const u64 es = d.home + (u64)fe;
        const u64 ann = SK ? stage_keys[(size_t)((u32)es & bmask) * W] : set.keys[es * W];
        sl = nb0 + KC_ANN_COUNT(ann) - (u64)(fe - j);   // (no match: j = 8 > fe, the value is not used)
    }
    return true;
}

// The two ports of one k-mer in one go (hash-sharded flow).  SH: a port whose home lies in another rank's slot range is not
// looked up: its four candidates come back as KC_REMOTE and the owner answers the queries later.
template <int W> struct KcCand2 {
    Key<W> x, xr;    // the k-mer and its reverse complement
    u64 cslot[8];    // candidate i = port (i >> 2), extension character (i & 3): its slot, KC_NONE64 (absent or not a legal edge), or KC_REMOTE
    u64 phome[2];    // the home the 4 candidates of port L (0) / R (1) share
    u32 qbits;       // bit i: the port on which candidate i is entered (set for the candidates that were found)
    bool xpal;       // x is its own reverse complement (links on port L only)
};
// x, RC(x) and the minimizer hashes of the two candidate middles: m-mer starts 2..k-m (append side) / 0..k-m-2 (prepend side) of x
template <int W, int KC>
__device__ __forceinline__ void kc_kmer_mins(const Key<W> &x, const Key<W> &xr, int k, int m, u32 &minL, u32 &minR) {
    // (the two windows share all but their two outer m-mers: one running minimum over the shared ones, the outer ones joined at the end)
    u32 both = ~0u, onlyL = ~0u, onlyR = ~0u;
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const int top = 2 * (m - 1);
    u32 f = 0, r = 0;
    if constexpr (W != 1) {
#pragma unroll
        for (int j = 0; j < m - 1; ++j) {
            u32 b = key_at(x, k, j);
            f = (f << 2) | b;
            r = (r >> 2) | ((3u - b) << top);
        }
    }
#pragma unroll
    for (int i = 0; i <= k - m; ++i) {
        if constexpr (W == 1) {   // one word: every m-mer is a shift of x / RC(x), no serial dependency between positions
            f = (u32)(x.w[0] >> (2 * (k - m - i))) & maskm;
            r = (u32)(xr.w[0] >> (2 * i)) & maskm;
        } else {
            u32 b = key_at(x, k, i + m - 1);
            f = ((f << 2) | b) & maskm;
            r = (r >> 2) | ((3u - b) << top);
        }
        const u32 h = kc_mmer_hash(f, r);
        if constexpr (W == 1) {
            if (i >= 2 && i <= k - m - 2) both = h < both ? h : both;
            else if (i < 2) { if (i <= k - m - 2) onlyL = h < onlyL ? h : onlyL; }
            else onlyR = h < onlyR ? h : onlyR;
        } else {
            // (wide keys: two plain running minima.  The three-way form above, one instruction per position cheaper, cost the 127-mer
            // kernel its register budget in round 5 -- 123 registers and no spill became 128 + 98 spilled, 7.6 -> 22 ms per 10^8
            // k-mers -- and nobody measured wide keys that round; found and undone in round 6)
            if (i >= 2) onlyR = h < onlyR ? h : onlyR;
            if (i <= k - m - 2) onlyL = h < onlyL ? h : onlyL;
        }
    }
    minL = both < onlyL ? both : onlyL;
    minR = both < onlyR ? both : onlyR;
}
template <int W, int KC, bool SH>
__device__ __forceinline__ void kc_cand_lookup_mid(const SetView<W> &set, int k_rt, int complements, const u64 slot, const u8 *stage_tags, const u64 sblk,
                                                   KcCand2<W> &out) {
    const int k = KC ? KC : k_rt;
    const int m = KC ? KC_MSPEC : set.m;
    const Key<W> x = set_key_at(set, slot);
    const Key<W> xr = key_rc(x, k);
    out.x = x; out.xr = xr;
    out.xpal = (KC & 1) ? false : (complements && key_eq(x, xr));   // odd k: no palindromes
    u32 minL, minR;
    kc_kmer_mins<W, KC>(x, xr, k, m, minL, minR);
    u32 qbits = 0;
#pragma unroll
    for (int port = 0; port < 2; ++port) {
        u64 cs[4] = {KC_NONE64, KC_NONE64, KC_NONE64, KC_NONE64};
        u32 qb = 0;
        const KcPortDesc d = kc_port_desc<W, KC>(set, x, xr, k, complements, port, port ? minR : minL);
        out.phome[port] = d.home;
        if (port == 1 && out.xpal) {
        } else if (SH && d.home - set.rlo >= set.rcap) {
#pragma unroll
            for (int c = 0; c < 4; ++c) cs[c] = KC_REMOTE;
        } else {
            bool ok;
            if (stage_tags && (d.home >> set.bshift) == sblk) ok = kc_port_search<W, true, false>(set, k, complements, x, xr, slot, port, d, stage_tags, nullptr, cs, qb);
            else ok = kc_port_search<W, false, false>(set, k, complements, x, xr, slot, port, d, nullptr, nullptr, cs, qb);
            if (!ok) kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out.cslot[port * 4 + c] = cs[c];
        qbits |= qb << (port * 4);
    }
    out.qbits = qbits;
}

// Round 1: one thread per k-mer handles both ports (they share x, RC(x), hash(x) and one pass
// over the m-mer hashes whose prefix/suffix minima give both ports' shared minimizer part).
// `stage` (LDS) holds the tag bytes of table block `sblk` (+ 8 bytes of the next one) when sblk != ~0: see the kernel below.
template <int W, int KC>
__device__ __forceinline__ void kc_choose_dense_one(const SetView<W> &set, u64 *__restrict__ link, int k_rt, int complements, const u64 slot,
                                                    const u64 *stage, const u64 sblk, u64 *heads, u64 heads_shard_cap, u64 *heads_cursors,
                                                    u64 *branch, u64 branch_shard_cap, u64 *branch_cursors) {
    KcCand<W> cd;
    kc_cand_lookup<W, KC, false>(set, k_rt, complements, slot, stage, sblk, cd);
    const Key<W> &x = cd.x;
    const Key<W> (&cy)[8] = cd.cy;
    const u64 (&cslot)[8] = cd.cslot;
    const u32 qbits = cd.qbits;
    const bool xpal = cd.xpal;
    // every 256th k-mer (by dense id) is a splitter: chains are walked segment-wise between splitters
    const u64 nid = set_nid_of_slot(set, slot);
    const u64 splitbit = ((nid & 0xffull) == 0) ? LK_SPLIT : 0;
#pragma unroll
    for (int port = 0; port < 2; ++port) {
        const u64 pid = (nid << 1) | (u64)port;
        if (port == 1 && xpal) {
            link[pid] = LK_MAKE(LK_DEAD, 0, 0, slot) | splitbit;
            if (complements) kc_push_sharded(pid, heads, heads_shard_cap, heads_cursors);
            continue;
        }
        int nfound = 0;
        Key<W> by = x; int bq = 0; u64 bpr = 0, bslot = 0, hxk = 0; u32 bbase = 0;
#pragma unroll
        for (u32 c = 0; c < 4; ++c) {
            const int i = port * 4 + (int)c;
            const u64 tslot = cslot[i];
            if (tslot == KC_NONE64) continue;
            const Key<W> y = cy[i];
            const int q = (int)((qbits >> i) & 1);
            if (nfound == 0) { by = y; bq = q; bslot = tslot; bbase = port ? c : 3 - c; nfound = 1; continue; }
            // priorities (the layout-free key hash) are only needed to choose between two existing candidates
            if (nfound == 1) { hxk = key_hash(x); bpr = kc_mix64(kc_port_term(hxk, port) + kc_port_term(key_hash(by), bq)); nfound = 2; }
            u64 pr = kc_mix64(kc_port_term(hxk, port) + kc_port_term(key_hash(y), q));
            if (edge_preferred(x, port, y, q, pr, by, bq, bpr)) { by = y; bq = q; bpr = pr; bslot = tslot; bbase = port ? c : 3 - c; }
        }
        link[pid] = (nfound ? LK_MAKE(LK_CHOICE, bq, bbase, set_nid_of_slot(set, bslot)) : LK_MAKE(LK_DEAD, 0, 0, slot)) | splitbit;
        if (!nfound && (complements || port == 0)) kc_push_sharded(pid, heads, heads_shard_cap, heads_cursors);   // cf. kc_push_head
        if (nfound >= 2) kc_push_sharded(pid | KC_WK_TAG(slot & (((u64)1 << set.bshift) - 1)), branch, branch_shard_cap, branch_cursors);   // branch port: it has to reject the others
    }
}

// The kernel: a workgroup owns KC_CHOOSE_SLOTS consecutive table slots (a quarter of an 8192-slot block) and first copies
// the tag bytes of the whole block into LDS.  ~90 % of the candidate lookups start in that very block (shared minimizer),
// so their tag windows are read from LDS instead of each dragging a 128-byte line through the L1: the kernel was bound by
// exactly that L2->L1 fill traffic (8 scattered tag lines per k-mer), not by HBM, ALU work or latency.  The occupied slots
// of its range are listed in LDS from the staged tags, so the lanes stay dense without any global list of occupied slots.
#define KC_CHOOSE_SLOTS 2048
template <int W, int KC>
__global__ void __launch_bounds__(256) kc_choose_dense_kernel(SetView<W> set, u64 *__restrict__ link, int k_rt, int complements, int xcd_aware,
                                                              u64 *heads, u64 heads_shard_cap, u64 *heads_cursors, u64 *branch,
                                                              u64 branch_shard_cap, u64 *branch_cursors, int stage_words) {
    extern __shared__ u64 kc_stage[];
    __shared__ unsigned short kc_list[KC_CHOOSE_SLOTS];
    __shared__ u32 kc_nlist;
    const u64 base = kc_xcd_block_id(xcd_aware) * (u64)KC_CHOOSE_SLOTS;
    if (base >= set.cap) return;
    if (threadIdx.x == 0) kc_nlist = 0;
    u64 sblk = ~(u64)0;
    if (stage_words) {
        sblk = base >> set.bshift;
        const u64 *src = reinterpret_cast<const u64 *>(set.tags + (sblk << set.bshift));
        for (int j = threadIdx.x; j < stage_words; j += 256) kc_stage[j] = src[j];
    }
    __syncthreads();
    // (blocks smaller than the slot range -- only with the KC_BSHIFT knob -- are staged for the window reads of the first
    // block only; the occupancy of the whole range then comes from global memory)
    const bool range_in_block = stage_words && ((u64)1 << set.bshift) >= (u64)KC_CHOOSE_SLOTS;
    const u8 *tg = range_in_block ? reinterpret_cast<const u8 *>(kc_stage) + (base & (((u64)1 << set.bshift) - 1)) : set.tags + base;
    for (u32 i = threadIdx.x; i < KC_CHOOSE_SLOTS; i += 256)
        if (base + i < set.cap && tg[i] != 0) kc_list[atomicAdd(&kc_nlist, 1u)] = (unsigned short)i;
    __syncthreads();
    const u32 n = kc_nlist;
#pragma unroll 1
    for (u32 t = threadIdx.x; t < n; t += 256)
        kc_choose_dense_one<W, KC>(set, link, k_rt, complements, base + kc_list[t], kc_stage, sblk, heads, heads_shard_cap, heads_cursors, branch,
                                   branch_shard_cap, branch_cursors);
}

// ---- round 1 on a table with middle-based homes: one workgroup per table block.  The block's tag bytes and (single-word keys)
// its keys are staged in LDS with coalesced 16-byte loads; ~90 % of the ports of the block's k-mers have their home in that
// very block, so they are settled out of LDS.  The block is swept in quarters: the occupied slots of a quarter are listed in
// LDS (so that the lanes stay dense), every listed k-mer settles its in-block ports and lists the others; the listed
// ("external") ports are then looked up in global memory with every lane busy -- handled inline they made each sweep step of
// every wave wait for two dependent global round trips on behalf of a few lanes (measured: waves waited 70 % of their cycles).
// LDS: tags + keys + the two lists = 76 KB for 8192-slot blocks of 64-bit keys, two workgroups per CU (80 KB each did not fit twice).
// `nid` = dense id of the k-mer in `slot`; cs[] (+ delta) = dense ids of the candidates.  A DEAD word carries the SLOT of its own k-mer:
// chain ends are where a key is needed (orientation, order, first k-mer), and the walks have that word in their hands there.
template <int W>
__device__ __forceinline__ u64 kc_choose_port(int k, int complements, const u64 nid, const u64 slot, const int port, const Key<W> &x,
                                               const Key<W> &xr, const u64 (&cs)[4], const u32 qb, u64 *heads, u64 heads_shard_cap, u64 *heads_cursors,
                                               u64 *branch, u64 branch_shard_cap, u64 *branch_cursors, const u64 delta = 0, const u64 wk_tag = 0) {
    // (cs[] + delta = the candidates' dense ids: the block kernel passes slots and the one difference its window search found)
    // every 256th k-mer (by id) is a splitter: chains are walked segment-wise between splitters
    const u64 splitbit = ((nid & 0xffull) == 0) ? LK_SPLIT : 0;
    const u64 pid = (nid << 1) | (u64)port;
    int nfound = 0;
    u32 bc = 0, bq = 0;
    u64 bslot = 0;
#pragma unroll
    for (u32 c = 0; c < 4; ++c) {
        if (cs[c] == KC_NONE64) continue;
        ++nfound; bc = c; bslot = cs[c]; bq = (qb >> c) & 1u;
    }
    if (nfound >= 2) {   // a branch port (rare): the best edge by the layout-free priority; the port will reject the others
        const u64 hxk = key_hash(x);
        Key<W> by = x; u64 bpr = 0; bool have = false;
        for (u32 c = 0; c < 4; ++c) {
            u64 ts = KC_NONE64;
#pragma unroll
            for (u32 cc = 0; cc < 4; ++cc) ts = (c == cc) ? cs[cc] : ts;
            if (ts == KC_NONE64) continue;
            int q;
            const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
            const u64 pr = kc_mix64(kc_port_term(hxk, port) + kc_port_term(key_hash(y), q));
            if (!have || edge_preferred(x, port, y, q, pr, by, (int)bq, bpr)) { by = y; bq = (u32)q; bpr = pr; bslot = ts; bc = c; have = true; }
        }
    }
    if (!nfound && (complements || port == 0)) kc_push_sharded(pid, heads, heads_shard_cap, heads_cursors);   // cf. kc_push_head
    if (nfound >= 2) kc_push_sharded(pid | wk_tag, branch, branch_shard_cap, branch_cursors);
    return (nfound ? LK_MAKE(LK_CHOICE, bq, port ? bc : 3 - bc, bslot + delta) : LK_MAKE(LK_DEAD, 0, 0, slot)) | splitbit;
}

// ---- lists of the hash-sharded flow's round 1 (kc_shard.hip): queries for the ports another rank answers, the edges of branch ports
#define KC_USH 16       /* cursor shards of the query list */
#define KC_USTRIDE 16   /* words between two cursors: one 128-byte line each */
struct ShLists {
    u64 *qry;        // queries {key[W], pid, home}: W + 2 words, shard s owns [s * qry_scap, ...)
    u64 qry_scap;
    u64 *qry_cur;    // KC_USH cursors
    u64 *edges;      // branch ports list their edges {x pid, y pid, priority, base, key x, key y}: 4 + 2 W words
    u64 edge_cap;
    u64 *edge_cur;
    u64 *flags;      // [0] a list overflowed
};

// Settle one port from the slots of its candidates (cs[c], entered on port bit c of qb): the link word.  A branch port (>= 2
// existing candidates) chooses its best edge here too -- provisionally -- and lists all its edges for the replicated branch
// matching, which overrides the choice where the sequential rule says so.
// (pid: the port by dense node id; cs[]: the candidates' dense node ids; a dead port's word is LK_DEAD here -- whoever writes it into the
// link array puts the slot of the port's k-mer in, which only the port's own rank knows)
template <int W>
__device__ __forceinline__ u64 kc_shard_settle_port(int k, int complements, const u64 pid, const Key<W> &x, const Key<W> &xr, const u64 (&cs)[4], const u32 qb,
                                                    const ShLists &L) {
    const int port = (int)(pid & 1);
    u32 nfound = 0, bc = 0;
#pragma unroll
    for (u32 c = 0; c < 4; ++c)
        if (cs[c] != KC_NONE64) { ++nfound; bc = c; }
    if (nfound == 0) return LK_DEAD;
    u64 bslot = 0;
#pragma unroll
    for (u32 c = 0; c < 4; ++c) bslot = (c == bc) ? cs[c] : bslot;
    if (nfound == 1) return LK_MAKE(LK_CHOICE, (qb >> bc) & 1u, port ? bc : 3 - bc, bslot);
    const u64 hx = key_hash(x);
    Key<W> by = x; int bq = 0; u64 bpr = 0; bool have = false;
    const u64 e0 = atomicAdd((ull *)L.edge_cur, (ull)nfound);
    if (e0 + nfound > L.edge_cap) L.flags[0] = 1;
    u32 ne = 0;
    for (u32 c = 0; c < 4; ++c) {
        u64 ts = KC_NONE64;
#pragma unroll
        for (u32 cc = 0; cc < 4; ++cc) ts = (c == cc) ? cs[cc] : ts;
        if (ts == KC_NONE64) continue;
        int q;
        const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
        const u64 pr = kc_mix64(kc_port_term(hx, port) + kc_port_term(key_hash(y), q));
        if (!have || edge_preferred(x, port, y, q, pr, by, bq, bpr)) { by = y; bq = q; bpr = pr; bc = c; bslot = ts; have = true; }
        if (e0 + ne < L.edge_cap) {
            u64 *e = L.edges + (e0 + ne) * (4 + 2 * W);
            e[0] = pid; e[1] = (ts << 1) | (u64)q; e[2] = pr; e[3] = port ? c : 3 - c;
            for (int w = 0; w < W; ++w) { e[4 + w] = x.w[w]; e[4 + W + w] = y.w[w]; }
        }
        ++ne;
    }
    return LK_MAKE(LK_CHOICE, bq, port ? bc : 3 - bc, bslot);   // provisional; the branch matching overrides it
}


#define KC_PROBE_SSTACK 128   /* per wave: occupied slots waiting for a full step (< 64 left over + 64 new) */
// per wave: listed ports waiting for a full step (< 64 left over + the new ones of one port; more go to the spill list).  MM: the
// minimizer hash of the port's middle is stacked with it (4 more bytes per entry); else the external pass recomputes it.
#define KC_PROBE_ESTACK(MM) ((MM) ? 120 : 112)
static inline size_t kc_probe_lds_bytes(int bshift, int W, bool stage_keys, int nt, bool mm) {
    const size_t bs = (size_t)1 << bshift;
    return bs + (stage_keys ? bs * W * 8 : 0) + (size_t)(nt / 64) * (KC_PROBE_SSTACK * 2 + KC_PROBE_ESTACK(mm) * (mm ? 6 : 2));
}
// SH (hash-sharded flow): the workgroups sweep this rank's blocks; a port whose home lies in another rank's range becomes ONE query
// {x, port id, home} (the owner answers with the finished link word); ports are settled by kc_shard_settle_port (plain link
// words, branch ports list their edges in L), nothing is listed for the single-GPU chain discovery.
// GP (round 6; only for tables all of whose keys sit at their PRIMARY home -- kc_set::homes_primary): the sweep does not compute a
// k-mer's minimizers at all.  It GUESSES that the candidates of a port live in the k-mer's own block (87 % do), searches the window
// at the in-block offset the middle's hash gives, and keeps the result only when a neighbour is FOUND there: a key found at a
// primary home inside this block proves that the port's home IS in this block, so the window held all candidates.  Anything else --
// nothing found (the home is in another block, or the port is dead), an unsettled window -- is listed, and the listed pass computes
// the one minimizer it needs, as it always did.  16 m-mer hashes per k-mer become 14 per listed port (0.17 ports per k-mer).
template <int W, int KC, bool SK, int NT, bool MM, bool SH = false, bool GP = false>
__global__ void __launch_bounds__(NT, NT == 1024 ? 8 : 4) kc_probe_block_kernel(SetView<W> set, u64 *__restrict__ link, int k_rt, int complements, int xcd_aware,
                                                            u64 *heads, u64 heads_shard_cap, u64 *heads_cursors, u64 *branch,
                                                            u64 branch_shard_cap, u64 *branch_cursors, u64 *spill, u64 spill_cap,
                                                            u64 *spill_cursor, ShLists L = ShLists{}) {
    constexpr int KC_PROBE_THREADS = NT, KC_PROBE_WAVES = NT / 64, ECAP = KC_PROBE_ESTACK(MM);
    // DN (single-GPU flow): link words are addressed by DENSE node ids and carry dense ids -- 16 bytes per K-MER, written
    // contiguously (the k-mers of a block are numbered in slot order from block * set.nstride on), not 16 bytes per slot of a table at
    // load 0.5 written in 64-byte pieces.  The hash-sharded flow (SH) numbers its ports the same way (nstride is the same on every
    // rank, so a node id says which rank holds it); its words also carry the slot offset of their own k-mer (LK_OWN).
    constexpr bool DN = true;
    // One workgroup per table block; after the staging barrier its wavefronts work on their own: wave w sweeps the slots
    // [w bs/8, (w+1) bs/8) in rows of 64, queues the occupied ones and takes 64 off the queue whenever that many wait, so
    // every step runs with all lanes busy and no wave ever waits for another (with block-wide lists and barriers every stretch
    // ended in a nearly empty step that the other seven waves sat out: 4.1 ms against 3.6 ms per 10^8 k-mers).  A port that the
    // one-window search does not settle (home in another block: ~10 %; second match or no empty slot in the window: ~5 %) goes on
    // the wave's port stack with the minimizer hash of its middle, and is looked up in global memory 64 at a time.
    // The slot queue is first-in first-out (a ring of 128): the j-th k-mer a wave takes is the j-th occupied slot of its share, so a
    // lane knows its dense id without looking anything up, and the 64 lanes of a step write 1 KB of consecutive link words.
    // What a wave has LEFT when its share is swept (< 64 k-mers, < 64 listed ports) is not worked off in two nearly empty steps per
    // wave (round 6: with 16 waves per block that was 32 half-empty steps of ~100 per workgroup): the leftovers of all waves are
    // handed over to ONE list of the workgroup -- each k-mer with its rank -- and taken from there 64 at a time by the waves in turn,
    // first the k-mers, then the listed ports (two barriers per hand-over; the lists live where the rings / the stacks were).
    // dynamic LDS: [tags: bs][keys: bs * W words, if SK][slot rings: u16 x 128 per wave][port stacks: u16 x ECAP per wave][MM: u32 x ECAP per wave]
    // Single-word keys: 1024 threads and port stacks without the hash = 79 KB, two workgroups per CU = 8 waves per SIMD (the kernel
    // waits on LDS and memory more than it issues: at 4 waves per SIMD its VALU pipes were half idle).
    extern __shared__ ulonglong2 kc_pb[];
    __shared__ u32 kc_wcnt[NT / 64], kc_wleft[2][NT / 64];
    const int k = KC ? KC : k_rt;
    const int m = KC ? KC_MSPEC : set.m;
    const u64 lblk = kc_xcd_block_id(xcd_aware & 1);   // (SH: index among this rank's blocks)
    if (lblk >= (SH ? set.rcap >> set.bshift : set.nblocks)) return;
    const u64 blk = (SH ? set.rlo >> set.bshift : 0) + lblk;
    const u32 bs = 1u << set.bshift;
    const u64 base = blk << set.bshift;
    u8 *st = reinterpret_cast<u8 *>(kc_pb);
    u64 *sk = reinterpret_cast<u64 *>(st + bs);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);   // (wv: wave-uniform, kept in a scalar register)
    u8 *lists = st + bs + (SK ? (size_t)bs * W * 8 : 0);
    unsigned short *sstack = reinterpret_cast<unsigned short *>(lists) + (size_t)wv * KC_PROBE_SSTACK;
    unsigned short *epid = reinterpret_cast<unsigned short *>(lists + (size_t)KC_PROBE_WAVES * KC_PROBE_SSTACK * 2) + (size_t)wv * ECAP;
    u32 *emmh = reinterpret_cast<u32 *>(lists + (size_t)KC_PROBE_WAVES * (KC_PROBE_SSTACK * 2 + ECAP * 2)) + (size_t)wv * ECAP;   // (MM only; 4-byte aligned: ECAP is even)
    // the workgroup's lists of leftovers: k-mers {slot offset, rank << 16} where the rings were, listed ports where the stacks were
    u32 *sh_k = reinterpret_cast<u32 *>(lists);                                                                       // WAVES x 64 entries
    unsigned short *sh_p = reinterpret_cast<unsigned short *>(lists + (size_t)KC_PROBE_WAVES * KC_PROBE_SSTACK * 2);   // WAVES x ECAP entries
    u32 *sh_m = reinterpret_cast<u32 *>(lists + (size_t)KC_PROBE_WAVES * (KC_PROBE_SSTACK * 2 + ECAP * 2));            // (MM: their minimizer hashes, where the waves' emmh were)
    {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(set.tags + base);
        for (u32 j = threadIdx.x; j < bs / 16; j += KC_PROBE_THREADS) kc_pb[j] = src[j];
        if (SK) {
            const ulonglong2 *ksrc = reinterpret_cast<const ulonglong2 *>(set.keys + base * W);
            ulonglong2 *kdst = reinterpret_cast<ulonglong2 *>(sk);
            for (u32 j = threadIdx.x; j < bs * W / 2; j += KC_PROBE_THREADS) kdst[j] = ksrc[j];
        }
    }
    __syncthreads();
    const u32 per_wave = bs / KC_PROBE_WAVES, rows = (per_wave + 63) / 64;
    // DN: the number of k-mers in the shares of the waves before this one (the dense id of this wave's first k-mer)
    u64 nb0 = 0;
    u32 wfirst = 0;
    if (DN) {
        u32 mine = 0;
        for (u32 row = 0; row < rows; ++row) {
            const u32 off = row * 64 + (u32)lane, so = (u32)wv * per_wave + off;
            mine += (u32)__popcll(__ballot(off < per_wave && st[so] != 0));
        }
        if (lane == 0) kc_wcnt[wv] = mine;
        __syncthreads();
        for (int i = 0; i < wv; ++i) wfirst += kc_wcnt[i];
        wfirst = __builtin_amdgcn_readfirstlane(wfirst);
        nb0 = blk * (u64)set.nstride;
    }
    auto load_x = [&](u32 so) {
        Key<W> r;
        if (SK) {
#pragma unroll
            for (int w = 0; w < W; ++w) r.w[w] = sk[(size_t)so * W + w];
        } else r = set_key_at(set, base + so);
        return r;
    };
    // DN: k-mers of this block before slot offset `so` (occupied), from the staged tags and the count in the next empty slot
    auto local_rank = [&](u32 so) -> u64 {
        const u64 hi7 = 0x8080808080808080ULL;
        u32 e = so + 1;
        while (e < bs) {
            u64 wd;
            __builtin_memcpy(&wd, st + e, 8);   // (the bytes behind the tags are LDS of this workgroup: masked below)
            u64 emp = ~wd & hi7;
            if (bs - e < 8) emp &= ((u64)1 << (8 * (bs - e))) - 1;
            if (emp) { e += (u32)__builtin_ctzll(emp) >> 3; break; }
            e += 8;
        }
        if (e >= bs) return (u64)set.ncount[blk] - (u64)(bs - so);
        const u64 ann = SK ? sk[(size_t)e * W] : set.keys[(base + e) * W];
        return KC_ANN_COUNT(ann) - (u64)(e - so);
    };
    auto to_spill = [&](u64 pid) {   // (by SLOT: the spill kernels find the dense id themselves)
        const u64 at = atomicAdd((ull *)spill_cursor, 1ULL);
        if (at < spill_cap) spill[at] = pid;
        else if (SH) L.flags[0] = 1;   // (the sharded flow repeats the sweep with larger lists)
    };
    const unsigned long long ltm = ((unsigned long long)1 << lane) - 1;
    auto wave_sync = [&]() {   // LDS traffic between the lanes of this wave: the hardware keeps a wave's LDS instructions in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // a listed port (called by all lanes of the wave that are in the current step; `on` = this lane has one): the window search
    // in global memory, or -- SH, home on another rank -- a query.  have_mm: mmh_in is the minimizer hash the sweep stacked (MM, not GP)
    auto settle_listed = [&](bool on, u32 so, int port, u32 mmh_in, bool have_mm) {
        const u64 slot = base + so;
        bool remote = false;
        Key<W> x = key_zero<W>(), xr = x;
        KcPortDesc d{};
        if (on) {
            x = load_x(so);
            xr = key_rc(x, k);
            u32 mmh = mmh_in;
            if (!MM || GP || !have_mm) {
                const Key<W> lm = key_lowmask<W>(2 * (k - 2));
                const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
                mmh = kc_mid_min<W, KC>(mid, midr, k, m);
            }
            d = kc_port_desc<W, KC>(set, x, xr, k, complements, port, mmh);
            remote = SH && d.home - set.rlo >= set.rcap;
            if (!remote) {
                u64 cs[4]; u32 qb;
                u32 eoff = 0;
                const u64 sblk = (xcd_aware & 2) ? ~(u64)0 : blk;   // (bit 1: test knob, table reads only)
                // (tried in round 6 and dropped: a home in another block looked up from its first four KEY words alone -- one line instead of
                // tag line + key line whenever the run ends inside them -- measured 73.2 ms against 71.1 for this form at 2.5 10^9 k-mers)
                bool ok, skipped = false;
#ifdef KC_EXPERIMENTS
                // KC_PROBE_DOWN_SPILL (timing experiment, results stay right): what the sweep costs without half of its cross-block lookups --
                // a port whose home lies in an EARLIER block is not looked up; the spill list is declared overflowed instead, so that the
                // generic kernel redoes round 1 afterwards
                if ((xcd_aware & 16) && (d.home >> set.bshift) < blk) skipped = true;   // (the overflow is declared once per workgroup, below)
#endif
                ok = skipped ? true : kc_port_search<W, 2, SK, DN>(set, k, complements, x, xr, slot, port, d, st, sk, cs, qb, sblk, &eoff);
                auto delta_of = [&]() -> u64 { return kc_port_delta<W, SK>(set, d.home, eoff, (d.home >> set.bshift) == sblk, sk); };
                if (skipped) { }
                else if (!ok) to_spill((slot << 1) | (u64)port);   // (rare: generic lookups in a kernel of their own keep this one lean)
                else if (SH) {
                    // at most one candidate: the word; a branch port is settled (and lists its edges) in the spill kernel
                    u32 nf = 0, bc = 0;
                    u64 bs1 = 0;
#pragma unroll
                    for (u32 c = 0; c < 4; ++c)
                        if (cs[c] != KC_NONE64) { ++nf; bc = c; bs1 = cs[c]; }
                    if (nf >= 2) to_spill((slot << 1) | (u64)port);
                    else {
                        const u64 nid = nb0 + local_rank(so);
                        const u64 tgt = nf ? bs1 + delta_of() : slot;
                        link[(nid << 1) | (u64)port] = (nf ? LK_MAKE(LK_CHOICE, (qb >> bc) & 1u, port ? bc : 3 - bc, tgt) : LK_MAKE(LK_DEAD, 0, 0, tgt)) | LK_OWN_BITS(so);
                    }
                }
                else {
                    const u64 delta = delta_of();
                    const u64 nid = nb0 + local_rank(so);
                    link[(nid << 1) | (u64)port] = kc_choose_port<W>(k, complements, nid, slot, port, x, xr, cs, qb, heads, heads_shard_cap, heads_cursors, branch,
                                                                     branch_shard_cap, branch_cursors, delta, KC_WK_TAG(so));
                }
            }
        }
        if (SH) {   // query space: one reservation per wave and call
            const unsigned long long rm = __ballot(remote);
            if (rm) {
                const u32 shard = ((u32)blockIdx.x * (u32)KC_PROBE_WAVES + (u32)wv) & (KC_USH - 1);
                u64 qbase = 0;
                const int leader = __builtin_ctzll(rm);
                if (lane == leader) qbase = atomicAdd((ull *)&L.qry_cur[shard * KC_USTRIDE], (ull)__popcll(rm));
                qbase = __shfl(qbase, leader, 64);
                if (remote) {
                    const u64 qpos = qbase + (u64)__popcll(rm & ltm);
                    if (qpos >= L.qry_scap) L.flags[0] = 1;
                    else {
                        u64 *qq = L.qry + ((u64)shard * L.qry_scap + qpos) * (W + 2);
                        for (int w = 0; w < W; ++w) qq[w] = x.w[w];
                        qq[W] = ((nb0 + local_rank(so)) << 1) | (u64)port | ((u64)so << 48);   // the asking port by node id; bits 48-62: the slot offset of its k-mer (kc_shard_fill_kernel)
                        qq[W + 1] = d.home;
                    }
                }
            }
        }
    };
#ifdef KC_EXPERIMENTS
    if ((xcd_aware & 16) && threadIdx.x == 0 && lblk == 0) *spill_cursor = spill_cap + 1;   // (KC_PROBE_DOWN_SPILL: the generic kernel redoes round 1)
#endif
    u32 shead = 0, stail = 0, ecnt = 0;   // slot ring: taken / queued so far; entries on the port stack (all wave-uniform)
    // phase 0: the wave sweeps its share; 1: the workgroup's left-over k-mers, chunk by chunk; 2: the workgroup's left-over listed ports
    u32 row = 0, phase = 0, chunk = (u32)wv, nk_sh = 0, np_sh = 0;
    // (xcd_aware bit 2, A/B knob KC_PROBE_NO_SHARE: every wave keeps its leftovers to itself -- a list of its own inside the shared
    // area, worked off in one nearly empty step each, as before round 6)
    const bool own_left = (xcd_aware & 4) != 0;
    u32 lbase_k = 0, lbase_p = 0;
#pragma unroll 1
    for (;;) {
        // what this wave does next (wave-uniform): 0 = a step of 64 k-mers, 1 = a step of 64 listed ports, 2 = queue a row of slots,
        // 3 = hand the leftovers over to the workgroup's list, 4 = done
        int act;
        if (ecnt >= 64) act = 1;
        else if (phase == 0) act = (stail - shead >= 64) ? 0 : (row < rows ? 2 : 3);
        else if (phase == 1) act = (chunk * 64u < nk_sh) ? 0 : 3;
        else act = (chunk * 64u < np_sh) ? 1 : 4;
        if (act == 4) break;
        if (act == 2) {
            const u32 off = row * 64 + (u32)lane, so = (u32)wv * per_wave + off;
            const bool occ = off < per_wave && st[so] != 0;
            const unsigned long long mk = __ballot(occ);
            if (occ) sstack[(stail + (u32)__popcll(mk & ltm)) & (KC_PROBE_SSTACK - 1)] = (unsigned short)so;
            stail += (u32)__popcll(mk);
            ++row;
            wave_sync();
            continue;
        }
        if (act == 3) {
            // every lane takes one left-over entry into a register, the waves agree on their places in the workgroup's list (which lies
            // where the rings / stacks were: nobody writes before everybody has read), then the entries go there
            const u32 cnt = phase == 0 ? stail - shead : ecnt;   // (< 64)
            u32 ent = 0, ent_m = 0;
            if ((u32)lane < cnt) {
                if (phase == 0) ent = (u32)sstack[(shead + (u32)lane) & (KC_PROBE_SSTACK - 1)] | ((wfirst + shead + (u32)lane) << 16);
                else { ent = epid[lane]; if (MM && !GP) ent_m = emmh[lane]; }   // (wide keys: the minimizer hash travels with its port -- recomputing it is a serial pass over the key)
            }
            if (lane == 0) kc_wleft[phase][wv] = cnt;
            __syncthreads();
            u32 at = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < KC_PROBE_WAVES; ++i) { const u32 c = kc_wleft[phase][i]; at += i < wv ? c : 0u; tot += c; }
            at = __builtin_amdgcn_readfirstlane(at); tot = __builtin_amdgcn_readfirstlane(tot);
            if (own_left) { at = (u32)wv * (phase == 0 ? 64u : (u32)ECAP); tot = cnt; }
            if ((u32)lane < cnt) {
                if (phase == 0) sh_k[at + (u32)lane] = ent;
                else { sh_p[at + (u32)lane] = (unsigned short)ent; if (MM && !GP) sh_m[at + (u32)lane] = ent_m; }
            }
            __syncthreads();
            if (phase == 0) { nk_sh = tot; shead = stail; lbase_k = own_left ? at : 0u; } else { np_sh = tot; ecnt = 0; lbase_p = own_left ? at : 0u; }
            ++phase;
            chunk = own_left ? 0u : (u32)wv;
            continue;
        }
        if (act == 1) {
            // ---- 64 listed ports (the last chunk of the workgroup's list: fewer): generic window search in global memory
            bool on = true, have_mm = false;
            u32 e, mm_in = 0;
            if (phase < 2) {
                ecnt -= 64;
                e = epid[ecnt + lane];
                if (MM && !GP) { mm_in = emmh[ecnt + lane]; have_mm = true; }
            } else {
                const u32 idx = chunk * 64u + (u32)lane;
                on = idx < np_sh;
                e = on ? sh_p[lbase_p + idx] : 0u;
                if (MM && !GP) { mm_in = on ? sh_m[lbase_p + idx] : 0u; have_mm = true; }
                chunk += (u32)KC_PROBE_WAVES;
            }
            settle_listed(on, e >> 1, (int)(e & 1u), mm_in, have_mm);
            wave_sync();
            continue;
        }
        // ---- 64 k-mers, oldest first (the last chunk of the workgroup's list: fewer)
        bool active = true;
        u32 so, rank;                                                            // slot offset in the block; (DN) this k-mer's number within the block
        if (phase == 0) {
            so = sstack[(shead + (u32)lane) & (KC_PROBE_SSTACK - 1)];
            rank = wfirst + shead + (u32)lane;
            shead += 64;
        } else {
            const u32 idx = chunk * 64u + (u32)lane;
            active = idx < nk_sh;
            const u32 ent = active ? sh_k[lbase_k + idx] : 0u;
            so = ent & 0xffffu; rank = ent >> 16;
            chunk += (u32)KC_PROBE_WAVES;
        }
        const u64 slot = base + so;
        auto own = [&]() -> u64 { return DN ? nb0 + (u64)rank : slot; };        // what a port of this k-mer is numbered by
        const Key<W> x = load_x(so);
        const Key<W> xr = key_rc(x, k);
        const bool xpal = (KC & 1) ? false : (complements && key_eq(x, xr));   // odd k: no palindromes
        const bool split = !SH && (((u32)nb0 + rank) & 0xffu) == 0;            // every 256th k-mer is a splitter of the segment-parallel walks
        const u64 ownbits = SH ? LK_OWN_BITS(so) : 0;                           // (hash-sharded flow: the words carry the slot offset of their k-mer)
        auto deadw = [&]() -> u64 { return LK_MAKE(LK_DEAD, 0, 0, slot) | ownbits; };   // (a dead port's word carries the slot of its k-mer)
        u32 minL = 0, minR = 0;
        if (!GP) kc_kmer_mins<W, KC>(x, xr, k, m, minL, minR);
        u64 wd0 = 0, wd1 = 0;
        u32 have = 0;
#pragma unroll 1
        for (int port = 0; port < 2; ++port) {
            bool defer = false, done = false;
            u64 word = 0;
            const u32 mmh = port ? minR : minL;
            if (active) {
                if (port == 1 && xpal) {   // palindromes link on port L only
                    word = deadw();
                    done = true;
                    if (!SH && complements) kc_push_sharded((own() << 1) | 1, heads, heads_shard_cap, heads_cursors);
                } else {
                    const KcPortDesc d = kc_port_desc<W, KC, GP>(set, x, xr, k, complements, port, mmh, base);
                    bool found; u32 c, q; u64 sl;
                    if ((GP || (d.home >> set.bshift) == blk) && kc_port_search_one<W, SK, DN>(set, k, complements, x, xr, slot, port, d, st, sk, found, c, q, sl, nb0) && (!GP || found)) {
                        word = found ? (LK_MAKE(LK_CHOICE, q, port ? c : 3u - c, sl) | ownbits) : deadw();
                        done = true;
                        if (!SH && !found && (complements || port == 0)) kc_push_sharded((own() << 1) | (u64)port, heads, heads_shard_cap, heads_cursors);   // cf. kc_push_head
                    } else defer = true;
                }
            }
            const unsigned long long dm = __ballot(defer);
            if (dm) {
                const u32 nd = (u32)__popcll(dm);
                if (ecnt + nd <= (u32)ECAP) {
                    if (defer) { const u32 at = ecnt + (u32)__popcll(dm & ltm); epid[at] = (unsigned short)((so << 1) | (u32)port); if (MM && !GP) emmh[at] = mmh; }
                    ecnt += nd;
                } else if (defer) to_spill((slot << 1) | (u64)port);   // (no room on the stack; SH: the spill kernel turns remote ones into queries)
            }
            if (done) { if (port) wd1 = word; else wd0 = word; have |= 1u << port; }
        }
        // both words of a k-mer in one 16-byte store whenever both ports were settled here
        if (split) { wd0 |= LK_SPLIT; wd1 |= LK_SPLIT; }
        u64 *lw = link + (own() << 1);
        if (have == 3u) *reinterpret_cast<ulonglong2 *>(lw) = make_ulonglong2(wd0, wd1);
        else if (have == 1u) lw[0] = wd0;
        else if (have == 2u) lw[1] = wd1;
        wave_sync();
    }
}
// the ports the block kernel handed on (rare): generic lookups, one thread per port, grid-stride over the device-side count
template <int W>
__global__ void __launch_bounds__(256) kc_probe_spill_kernel(SetView<W> set, u64 *__restrict__ link, int k, int complements, const u64 *__restrict__ spill,
                                                             u64 spill_cap, const u64 *__restrict__ spill_cursor, u64 *heads, u64 heads_shard_cap,
                                                             u64 *heads_cursors, u64 *branch, u64 branch_shard_cap, u64 *branch_cursors) {
    u64 n = *spill_cursor;
    if (n > spill_cap) n = spill_cap;   // (overflow: the caller redoes round 1 with the generic kernel)
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (u64)gridDim.x * blockDim.x) {
        const u64 pid = spill[t], slot = pid >> 1;   // (listed by slot)
        const int port = (int)(pid & 1);
        const Key<W> x = set_key_at(set, slot);
        const Key<W> xr = key_rc(x, k);
        u64 cs[4]; u32 qb;
        kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cs[c] != KC_NONE64) cs[c] = set_nid_of_slot(set, cs[c]);
        const u64 nid = set_nid_of_slot(set, slot);
        link[(nid << 1) | (u64)port] = kc_choose_port<W>(k, complements, nid, slot, port, x, xr, cs, qb, heads, heads_shard_cap, heads_cursors, branch, branch_shard_cap, branch_cursors, 0,
                                                         KC_WK_TAG(slot & (((u64)1 << set.bshift) - 1)));
    }
}

// the same for the hash-sharded flow (a spilled port whose home lies on another rank becomes a query here)
template <int W>
__global__ void __launch_bounds__(256) kc_shard_spill_kernel(SetView<W> set, u64 *__restrict__ link, int k, int complements, const u64 *__restrict__ spill,
                                                             u64 spill_cap, const u64 *__restrict__ spill_cursor, ShLists L) {
    u64 n = *spill_cursor;
    if (n > spill_cap) n = spill_cap;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (u64)gridDim.x * blockDim.x) {
        const u64 spid = spill[t], slot = spid >> 1;   // (listed by slot)
        const int port = (int)(spid & 1);
        const Key<W> x = set_key_at(set, slot);
        const Key<W> xr = key_rc(x, k);
        const u64 pid = (set_nid_of_slot(set, slot) << 1) | (u64)port;
        const u64 so = slot & (((u64)1 << set.bshift) - 1);
        if (set.m) {
            const Key<W> lm = key_lowmask<W>(2 * (k - 2));
            const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
            const KcPortDesc d = kc_port_desc<W, 0>(set, x, xr, k, complements, port, kc_mid_min<W, 0>(mid, midr, k, set.m));
            if (d.home - set.rlo >= set.rcap) {
                const u32 shard = (u32)t & (KC_USH - 1);
                const u64 qpos = atomicAdd((ull *)&L.qry_cur[shard * KC_USTRIDE], 1ULL);
                if (qpos >= L.qry_scap) L.flags[0] = 1;
                else {
                    u64 *qq = L.qry + ((u64)shard * L.qry_scap + qpos) * (W + 2);
                    for (int w = 0; w < W; ++w) qq[w] = x.w[w];
                    qq[W] = pid | (so << 48);
                    qq[W + 1] = d.home;
                }
                continue;
            }
        }
        u64 cs[4]; u32 qb;
        kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cs[c] != KC_NONE64) cs[c] = set_nid_of_slot(set, cs[c]);
        u64 w = kc_shard_settle_port<W>(k, complements, pid, x, xr, cs, qb, L);
        if (LK_STATE(w) == LK_DEAD) w = LK_MAKE(LK_DEAD, 0, 0, slot);
        link[pid] = w | LK_OWN_BITS(so);
    }
}
// round 1 by generic lookups over the whole table, one thread per slot: the fallback when the spill list of the block kernel
// overflows (tables with many overfull blocks)
template <int W>
__global__ void __launch_bounds__(256) kc_probe_generic_kernel(SetView<W> set, u64 *__restrict__ link, int k, int complements, u64 *heads, u64 heads_shard_cap,
                                                               u64 *heads_cursors, u64 *branch, u64 branch_shard_cap, u64 *branch_cursors) {
    const u64 slot = kc_gtid();
    if (slot >= set.cap || set.tags[slot] == 0) return;
    const Key<W> x = set_key_at(set, slot);
    const Key<W> xr = key_rc(x, k);
    const bool xpal = complements && key_eq(x, xr);
    const u64 nid = set_nid_of_slot(set, slot);
    for (int port = 0; port < 2; ++port) {
        if (port == 1 && xpal) {
            link[(nid << 1) | 1] = LK_MAKE(LK_DEAD, 0, 0, slot) | (((nid & 0xffull) == 0) ? LK_SPLIT : 0);
            if (complements) kc_push_sharded((nid << 1) | 1, heads, heads_shard_cap, heads_cursors);
            continue;
        }
        u64 cs[4]; u32 qb;
        kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (cs[c] != KC_NONE64) cs[c] = set_nid_of_slot(set, cs[c]);
        link[(nid << 1) | (u64)port] = kc_choose_port<W>(k, complements, nid, slot, port, x, xr, cs, qb, heads, heads_shard_cap, heads_cursors, branch, branch_shard_cap, branch_cursors, 0,
                                                         KC_WK_TAG(slot & (((u64)1 << set.bshift) - 1)));
    }
}

#define KC_CHUNK_WORDS 32
#define KC_NO_CHUNK 0xffffffffu
struct KcRec {
    u64 *pool;          // chunks of KC_CHUNK_WORDS words
    u32 *chunk_next;    // next chunk of the same walk
    u64 pool_cap;       // chunks
    u64 *pool_cursor;
    u64 *chain_stream;  // per chain: first chunk (KC_NO_CHUNK if it has a single node)
    u64 *chain_tail;    // per chain: oriented last node (slot << 1 | orientation)
};

// ---- Where a cycle is opened (oracle: simplitigs_canonical_order).  The reference's delete-as-you-go walk never closes a cycle
// (simplitigs.h:161-183): it would have run on from one of the cycle's k-mers into a neighbouring chain.  So the candidate edges
// from a port x.p on a cycle to a port y.q that the matching left DEAD are taken in the matching's own order (priority, then the
// ordered endpoint tuple): if x's cycle is still closed and y.q still free, the cycle is opened at x.p and x.p is matched with
// y.q -- the cycle becomes the continuation of that chain, as in the reference's walk; the port that faced x.p dies.  A cycle
// that no such edge opens is cut at its minimum key on port L.  Ports that die here do not count as DEAD for this rule (LK_CUT):
// it depends on the matching only.  The k-mers on cycles are few: their rows and candidate edges come to the host, which walks
// the cycles and decides; the same code serves the hash-sharded flow.
// row of a k-mer on a cycle: {slot, link word of port L, link word of port R, key[W]}
// (DN: the k-mers are listed, and the ports numbered, by dense node id -- the single-GPU flow; else by slot)
template <int W, bool DN = false>
__global__ void __launch_bounds__(256) kc_cycle_rows_kernel(SetView<W> set, const u64 *__restrict__ link, const u64 *__restrict__ list, u64 n, u64 *rows) {
    const u64 t = kc_gtid();
    if (t >= n) return;
    const u64 id = list[t];
    const u64 slot = DN ? set_slot_of_nid(set, id) : id;
    u64 *r = rows + t * (3 + W);
    r[0] = id; r[1] = link[id << 1]; r[2] = link[(id << 1) | 1];
    const Key<W> key = set_key_at(set, slot);
    for (int w = 0; w < W; ++w) r[3 + w] = key.w[w];
}
// the base a walk appends when it steps into node y through port q (the LK_BASE of the port word that points there)
template <int W> __device__ __forceinline__ u64 kc_step_base(const Key<W> &y, int q, int k) { return q == 0 ? (y.w[0] & 3) : 3 - (u64)key_at(y, k, 0); }
// one candidate edge x.p -> y.q (y.q dead): {priority, x pid, y pid, base into y | base into x << 2, key y[W]}
template <int W>
__device__ __forceinline__ void kc_cycle_cand_emit(const Key<W> &x, u64 hx, u64 xslot, int port, const Key<W> &y, u64 yslot, int q, int k, u64 *cand, u64 cand_cap, u64 *cursor) {
    const u64 at = atomicAdd((ull *)cursor, 1ULL);
    if (at >= cand_cap) return;
    u64 *r = cand + at * (4 + W);
    r[0] = kc_mix64(kc_port_term(hx, port) + kc_port_term(key_hash(y), q));
    r[1] = (xslot << 1) | (u64)port; r[2] = (yslot << 1) | (u64)q;
    r[3] = kc_step_base(y, q, k) | (kc_step_base(x, port, k) << 2);
    for (int w = 0; w < W; ++w) r[4 + w] = y.w[w];
}
template <int W>
__global__ void __launch_bounds__(256) kc_cycle_cand_kernel(SetView<W> set, const u64 *__restrict__ link, int k, int complements, const u64 *__restrict__ list, u64 n,
                                                            u64 *cand, u64 cand_cap, u64 *cursor) {
    const u64 t = kc_gtid();
    if (t >= 2 * n) return;
    const u64 nid = list[t >> 1];   // (single-GPU flow: dense node ids)
    const int port = (int)(t & 1);
    const Key<W> x = set_key_at(set, set_slot_of_nid(set, nid));
    const Key<W> xr = key_rc(x, k);
    const u64 hx = key_hash(x);
    for (u32 c = 0; c < 4; ++c) {
        int q;
        const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
        if (key_eq(y, x)) continue;
        const u64 ys = set_find(set, y);
        if (ys == KC_NONE64) continue;
        const u64 yn = set_nid_of_slot(set, ys);
        const u64 yw = link[(yn << 1) | (u64)q];
        if (LK_STATE(yw) != LK_DEAD || (yw & LK_CUT)) continue;
        kc_cycle_cand_emit<W>(x, hx, nid, port, y, yn, q, k, cand, cand_cap, cursor);
    }
}
// write the decided link words {pid, word} (single-GPU flow: the splitter flag of the port stays; a port that dies gets the slot of
// its k-mer into its word, like every dead port)
template <int W>
__global__ void __launch_bounds__(256) kc_cycle_fix_kernel(SetView<W> set, u64 *link, const u64 *__restrict__ fix, u64 n) {
    const u64 t = kc_gtid();
    if (t >= n) return;
    const u64 pid = fix[2 * t];
    u64 w = fix[2 * t + 1];
    if (LK_STATE(w) == LK_DEAD) w |= set_slot_of_nid(set, pid >> 1) << 6;
    link[pid] = w | (link[pid] & LK_SPLIT);
}
// the head list after the cycles were opened: the old heads that are still dead (one that a cycle was attached to is not) and the
// ports that died.  A dead port is the entry of a directed chain (unidirectional sets: the L ports only, cf. kc_push_head).
static __global__ void __launch_bounds__(256) kc_heads_refresh_kernel(const u64 *__restrict__ link, const u64 *__restrict__ heads, u64 nheads, const u64 *__restrict__ fix, u64 nfix,
                                                                      int complements, u64 *out, u64 *cursor) {
    const u64 t = kc_gtid();
    if (t >= nheads + nfix) return;
    u64 pid;
    if (t < nheads) { pid = heads[t]; if (LK_STATE(link[pid]) != LK_DEAD) return; }
    else { pid = fix[2 * (t - nheads)]; if (LK_STATE(fix[2 * (t - nheads) + 1]) != LK_DEAD || !(complements || !(pid & 1))) return; }
    out[atomicAdd((ull *)cursor, 1ULL)] = pid;
}
// host: walk the cycles of `rows` (n x (3 + W)), take the candidate edges `cand` (m x (4 + W)) in order of preference; fix = {pid, link
// word} of every port that changes; returns the number of cycles opened
template <int W>
static inline u64 kc_cycles_decide(const std::vector<u64> &rows, u64 n, const std::vector<u64> &cand, u64 m, std::vector<u64> &fix) {
    const int RW = 3 + W, CW = 4 + W;
    auto key_lt_h = [](const u64 *a, const u64 *b) { for (int i = W - 1; i >= 0; --i) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; } return false; };
    auto key_eq_h = [](const u64 *a, const u64 *b) { for (int i = 0; i < W; ++i) if (a[i] != b[i]) return false; return true; };
    std::unordered_map<u64, u64> at;
    at.reserve(n * 2);
    for (u64 i = 0; i < n; ++i) at[rows[i * RW]] = i;
    std::vector<u64> cyc(n, ~(u64)0), minnode;
    for (u64 i = 0; i < n; ++i) {
        if (cyc[i] != ~(u64)0 || at[rows[i * RW]] != i) continue;   // (a k-mer listed twice: one row speaks for it)
        const u64 id = minnode.size();
        u64 cur = i, best = i, steps = 0;
        int co = 0;
        bool closed = false;
        while (steps++ <= n) {
            cyc[cur] = id;
            if (key_lt_h(&rows[cur * RW + 3], &rows[best * RW + 3])) best = cur;
            const u64 w = rows[cur * RW + (co == 0 ? 2 : 1)];   // leave through R when entered on L
            auto it = at.find(LK_SLOT(w));
            if (!LK_IS_MATCHED(w) || it == at.end()) break;     // (not a closed cycle of listed k-mers: left alone)
            cur = it->second;
            co = LK_PORT(w) == 0 ? 0 : 1;
            if (cur == i) { closed = true; break; }
        }
        minnode.push_back(closed ? best : ~(u64)0);
    }
    // the candidate edges in the matching's order of preference
    struct Edge { u64 pr; const u64 *ka, *kb; int pa, pb; const u64 *c; u64 id; };
    std::vector<Edge> edges;
    edges.reserve(m);
    for (u64 j = 0; j < m; ++j) {
        const u64 *c = &cand[j * CW];
        auto it = at.find(c[1] >> 1);
        if (it == at.end()) continue;
        const u64 id = cyc[it->second];
        if (id == ~(u64)0 || minnode[id] == ~(u64)0) continue;
        if (at.count(c[2] >> 1)) continue;   // (a dead port is never on a cycle; kept for safety)
        const u64 *kx = &rows[it->second * RW + 3], *ky = c + 4;
        const int px = (int)(c[1] & 1), py = (int)(c[2] & 1);
        const bool x_first = key_lt_h(kx, ky) || (key_eq_h(kx, ky) && px <= py);
        edges.push_back(Edge{c[0], x_first ? kx : ky, x_first ? ky : kx, x_first ? px : py, x_first ? py : px, c, id});
    }
    std::sort(edges.begin(), edges.end(), [&](const Edge &a, const Edge &b) {
        if (a.pr != b.pr) return a.pr > b.pr;
        if (!key_eq_h(a.ka, b.ka)) return key_lt_h(a.ka, b.ka);
        if (a.pa != b.pa) return a.pa < b.pa;
        if (!key_eq_h(a.kb, b.kb)) return key_lt_h(a.kb, b.kb);
        return a.pb < b.pb;
    });
    std::vector<u8> opened(minnode.size(), 0);
    std::unordered_set<u64> taken;
    fix.clear();
    u64 ncyc = 0;
    auto kill_partner = [&](u64 pid) {
        const u64 partner = LK_PID(rows[at[pid >> 1] * RW + 1 + (pid & 1)]);   // the port that faces it
        fix.push_back(partner); fix.push_back(LK_DEAD | LK_CUT);
    };
    for (const Edge &e : edges) {
        const u64 xpid = e.c[1], ypid = e.c[2];
        if (opened[e.id] || taken.count(ypid)) continue;
        opened[e.id] = 1; taken.insert(ypid); ++ncyc;
        kill_partner(xpid);
        fix.push_back(xpid); fix.push_back(LK_MAKE(LK_MATCHED, ypid & 1, e.c[3] & 3, ypid >> 1));
        fix.push_back(ypid); fix.push_back(LK_MAKE(LK_MATCHED, xpid & 1, (e.c[3] >> 2) & 3, xpid >> 1));
    }
    for (size_t id = 0; id < minnode.size(); ++id) {
        if (minnode[id] == ~(u64)0 || opened[id]) continue;
        const u64 pid = rows[minnode[id] * RW] << 1;
        kill_partner(pid);
        if (LK_PID(rows[minnode[id] * RW + 1]) != pid) { fix.push_back(pid); fix.push_back(LK_DEAD | LK_CUT); }   // (a port matched with itself: one word)
        ++ncyc;
    }
    return ncyc;
}

// ---------------------------------------------------------------------------------------------------------
// Fragment contraction.  A chain walk costs one scattered 128-byte line per step for 8 useful bytes (measured: the walk
// kernel moves 16x its algorithmic bytes and sits at the HBM transaction ceiling), but ~90 % of its steps stay inside one
// table block (consecutive k-mers share their minimizer).  So every block first contracts the chain pieces that lie inside
// it: one workgroup loads the block's port links (16 B per slot) into LDS, walks every in-block run ("fragment") there from
// both of its ends, and replaces the INWARD link word of each end node by a shortcut: far end node, the number of nodes, and
// the appended bases of the run (up to 19 inline, longer runs through a word pool).  The chain walk then hops from fragment
// to fragment: two scattered lines per fragment instead of one per node.  Every replaced word is logged first (8 B per
// fragment end): if the hop walk finds a chain beyond the limit or misses k-mers (cycles), kc_frag_restore_kernel puts the
// words back and the classic flow takes over on the original links; inputs whose chains are long on average skip the
// contraction in the first place.  An end that finds no room in a list or pool simply keeps its plain link (the walk then
// steps through that fragment node by node from this side).
//
// MEASURED (MI355X, k = 31, 2.5e9 k-mers): contraction 43 ms + hop walk 32 ms against the plain walk's 111 ms.  (The hop walk
// first ran at 71 ms -- not for its memory accesses: scattered 8-byte reads run at ~49 G/s on this GPU whatever the footprint,
// tools/rand_probe.hip -- but because every walk reserved its first recording chunk with an atomic on one shared cursor and,
// once walks hop over runs of different lengths, the lanes of a wave reach that point in different loop trips, so the
// compiler's wave-level aggregation of the atomic no longer applied: ~5e6 same-address atomics at ~13 ns each.  The first
// chunk of walk t is now chunk t.)
//
// shortcut word: bits 0-1 = 0 (LK_OPEN, not a state of a finished port) and bit 5 (LK_REJ) set; bit 2: the bases are in the
// pool; bits 6-19 far node (slot offset in the block); bit 20 orientation at the far node; inline: bits 21-25 number of bases,
// bits 26-63 the bases (first one in the highest pair); pool: bits 21-34 nodes - 1, bits 35-63 first pool word (relative to
// the block's pool base).
#define KC_SC_IS(w) ((((w) & 3ull) == 0ull) && ((w) & LK_REJ))
#define KC_SC_EXT 4ull
#define KC_SC_NEXT 8ull
#define KC_SC_INLINE_MAX 19
#define KC_SC_NEXT_MAX 7
#define KC_FRAG_LONGS 512       /* runs of more than 20 nodes per block that get pool space (expected ~100) */
#define KC_FRAG_ENTRIES 4096    /* fragment ends per block (expected ~900 of 16384 ports) */
// DN (single-GPU flow): the links are addressed by dense node ids, so the block's link words are the contiguous range
// [blk * nstride, blk * nstride + ncount[blk]) x 16 bytes -- no tag bytes are read and no empty slot's words are streamed.  A shortcut then names
// its far node by the signed distance in ids (bits 6-19, two's complement) and a pool run by its absolute first word (bits 35-63),
// so the hop walk needs no block arithmetic.
template <int W, bool DN = false>
__global__ void __launch_bounds__(1024) kc_frag_contract_kernel(SetView<W> set, u64 *link, int complements, u64 *pool, u64 pool_cap, u64 *pool_cursors,
                                                                u64 *blk_pool_base, u64 *log, u64 log_cap, u64 *log_cursors, u64 *blk_log,
                                                                u64 *ent_out = nullptr, u64 ent_cap = 0, u64 *ent_cursor = nullptr, int no_next = 0, int only_unlogged = 0) {
    // only_unlogged: a second round over the blocks whose log is empty -- the ones kc_frag_unvisited_kernel put back to plain links for the
    // cycle rule (and those that never had anything to contract); the others keep their shortcuts.
    if (only_unlogged && blk_log[blockIdx.x] != 0) return;
    // no_next: no shortcut skips the far end of its run (the segment-parallel hop walks stop AT certain run ends, so every end must
    // be landed on).
    // ent_out (hash-sharded flow): the block's ENTRIES are listed on the way -- oriented nodes (slot << 1 | o) whose entry port o is
    // dead or matched to a slot of another rank, i.e. the starts of this rank's chain fragments (the link words are in registers
    // here anyway; a sweep of their own over 16 bytes per slot cost 15 ms per 10^9 k-mers).  One reservation per block.
    // LDS: per port of the block one 32-bit word -- bit 31: matched to a node of this block; then bits 0-13 that node's offset,
    // bit 14 the port entered there, bits 15-16 the appended base -- then the list of long runs and the list of fragment ends.
    // (76 KB for 8192-slot blocks: two workgroups share a CU.)
    extern __shared__ u32 kc_fl[];
    __shared__ u32 n_long, n_words, n_ent;
    const u32 bs = 1u << set.bshift;
    const u64 blk = (set.rlo >> set.bshift) + blockIdx.x;   // (hash-sharded tables: the blocks of this rank; pool / log arrays are indexed by the LOCAL block)
    const u64 base = DN ? blk * (u64)set.nstride : blk << set.bshift;               // first node of the block (DN: dense id, else slot)
    const u32 nn = DN ? set.ncount[blk] : bs;                                       // nodes (DN) / slots of the block
    const u64 lblk = blockIdx.x;
    u32 *longs = kc_fl + 2 * (size_t)bs;                            // (local port id of the entry, first pool word)
    const u32 long_cap = KC_FRAG_LONGS, fe_cap = KC_FRAG_ENTRIES;
    unsigned short *ents = reinterpret_cast<unsigned short *>(longs + 2 * long_cap);   // local port ids of the fragment ends
    if (threadIdx.x == 0) { n_long = 0; n_words = 0; n_ent = 0; }
    // one 16-byte load per node (both port words) and (slot-addressed links) one tag byte; bit 30 of the digest = occupied
    u32 emask = 0;   // bit 2 i + p: port p of the slot of sweep step i is an entry
    int step = 0;
    for (u32 s = threadIdx.x; s < nn; s += blockDim.x, ++step) {
        const bool occ = DN ? true : set.tags[base + s] != 0;
        const ulonglong2 ww = *reinterpret_cast<const ulonglong2 *>(link + (base + s) * 2);
        const u64 w2[2] = {ww.x, ww.y};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const u64 w = w2[p];
            const bool in = occ && LK_IS_MATCHED(w) && (DN ? LK_SLOT(w) - base < (u64)nn : (LK_SLOT(w) >> set.bshift) == blk);
            // (bits 17-18: state MATCHED rather than CHOICE, splitter flag -- with them the word of an in-block link can be rebuilt
            // exactly from the digest, so the log below needs no second, scattered read of the link array)
            kc_fl[2 * s + p] = (occ ? 0x40000000u : 0u) |
                               (in ? (0x80000000u | (u32)(LK_SLOT(w) - base) | ((u32)LK_PORT(w) << 14) | ((u32)LK_BASE(w) << 15) |
                                      (LK_STATE(w) == LK_MATCHED ? 1u << 17 : 0u) | ((w & LK_SPLIT) ? 1u << 18 : 0u)) : 0u);
            // (DN: node ids of this rank = [blocks before it * nstride, + its blocks * nstride))
            if (ent_out && occ && (p == 0 || complements) && step < 16 &&
                (LK_STATE(w) == LK_DEAD || (LK_IS_MATCHED(w) && (DN ? LK_SLOT(w) - (set.rlo >> set.bshift) * (u64)set.nstride >= (set.rcap >> set.bshift) * (u64)set.nstride
                                                                     : LK_SLOT(w) - set.rlo >= set.rcap))))
                emask |= 1u << (2 * step + p);
        }
    }
    __syncthreads();
    if (ent_out) {   // (uniform: every thread of the workgroup takes part)
        __shared__ u32 e_wsum[16];
        __shared__ u64 e_base;
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const u32 cnt = (u32)__popc(emask);
        u32 inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, d, 64); if (ln >= d) inc += o; }
        if (ln == 63) e_wsum[wv] = inc;
        __syncthreads();
        u32 before = 0, tot = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { if (i < wv) before += e_wsum[i]; tot += e_wsum[i]; }
        if (threadIdx.x == 0) e_base = tot ? atomicAdd((ull *)ent_cursor, (ull)tot) : 0;
        __syncthreads();
        u64 pos = e_base + before + inc - cnt;
        for (int j = 0; emask != 0 && j < 32; ++j)
            if ((emask >> j) & 1u) {
                if (pos < ent_cap) ent_out[pos] = ((base + (u64)threadIdx.x + (u64)(j >> 1) * blockDim.x) << 1) | (u64)(j & 1);
                ++pos;
            }
    }
    auto walk = [&](u32 s, int o, u32 &far, int &fo, u32 &len, u64 &bases, u64 *ext, bool fill) {
        // from the oriented node (s, o) along the links that stay in this block; bases = the appended bases, first one highest
        u32 cur = s;
        int co = o;
        len = 1; bases = 0;
        u64 acc = 0; int nacc = 0; u32 wi = 0;
        while (len <= bs) {   // (a run has at most bs nodes; the bound only guards against a corrupt table)
            const u32 w = kc_fl[2 * cur + (co == 0 ? 1 : 0)];
            if (!(w >> 31)) break;
            const u64 b = (w >> 15) & 3;
            if (len <= KC_SC_INLINE_MAX) bases |= b << (36 - 2 * (len - 1));
            if (fill) { acc = (acc << 2) | b; if (++nacc == 32) { ext[wi++] = acc; acc = 0; nacc = 0; } }
            cur = w & 0x3fffu;
            co = (int)((w >> 14) & 1);
            ++len;
        }
        if (fill && nacc) ext[wi] = acc << (64 - 2 * nacc);
        far = cur; fo = co;
    };
    // the far node as a shortcut names it: its offset in the block, or (DN) its signed distance from the end node s
    auto far_field = [&](u32 far, u32 s) -> u64 { return DN ? (u64)((far - s) & 0x3fffu) : (u64)far; };
    // The fragment ends first go into a list (one LDS atomic per wavefront and sweep step: ballot + popcount), so that the walks
    // below run with every lane busy -- walking straight from the slot sweep kept ~3 of 64 lanes active.
    const int lane = threadIdx.x & 63;
    for (u32 s0 = 0; s0 < nn; s0 += blockDim.x) {
        const u32 s = s0 + threadIdx.x;
        for (int o = 0; o < (complements ? 2 : 1); ++o) {
            // an end of a run of >= 2 nodes: occupied, not entered from inside the block, left towards a slot of the block
            const u32 f_in = s < nn ? kc_fl[2 * s + o] : 0u, f_out = s < nn ? kc_fl[2 * s + (o == 0 ? 1 : 0)] : 0u;   // (blocks of < 1024 slots: KC_BSHIFT)
            const bool is_end = (f_in & 0x40000000u) && !(f_in >> 31) && (f_out >> 31);
            const unsigned long long m = __ballot(is_end);
            if (!m) continue;
            u32 at0 = 0;
            if (lane == 0) at0 = atomicAdd(&n_ent, (u32)__popcll(m));
            at0 = __shfl(at0, 0, 64);
            if (is_end) {
                const u32 at = at0 + (u32)__popcll(m & (((unsigned long long)1 << lane) - 1));
                if (at < fe_cap) ents[at] = (unsigned short)(2 * s + (u32)o);   // (an end beyond the list keeps its plain link)
            }
        }
    }
    __syncthreads();
    const u32 ne = n_ent < fe_cap ? n_ent : fe_cap;
    // every word about to be replaced is logged first (kc_frag_restore_kernel puts it back if the hop walk has to be abandoned):
    // the original word with the local port id in its unused bits 46-59.  No log space: the block keeps its plain links.
    // (every block owns log_cap / gridDim.x log entries and pool_cap / gridDim.x pool words: a reservation from shared cursors
    // was a global atomic round trip that the whole workgroup waited for, twice per block)
    const u64 lcap_blk = log_cap / gridDim.x;
    const bool log_ok = ne <= lcap_blk;
    const u64 lbase = lblk * lcap_blk;
    if (threadIdx.x == 0) blk_log[lblk] = log_ok ? ((lbase << 16) | (u64)ne) : 0;
    if (!log_ok) return;
    for (u32 e = threadIdx.x; e < ne; e += blockDim.x) {
        const u32 s = ents[e] >> 1;
        const int o = (int)(ents[e] & 1);
        u32 far, len; int fo; u64 bases;
        walk(s, o, far, fo, len, bases, nullptr, false);
        const u32 xp = o == 0 ? 1u : 0u;
        {
            const u32 dg = kc_fl[2 * s + xp];   // (in-block by construction: the run continues through this port)
            const u64 orig = LK_MAKE((dg >> 17) & 1u ? LK_MATCHED : LK_CHOICE, (dg >> 14) & 1u, (dg >> 15) & 3u, base + (u64)(dg & 0x3fffu)) | ((dg >> 18) & 1u ? LK_SPLIT : 0ull);
            log[lbase + e] = orig | ((u64)(2 * s + xp) << 46);
        }
        // runs of up to 8 nodes whose far end is matched onwards also carry that onward link (the walk then skips the far end's
        // line altogether): bit 3 set, bits 6-41 next node, bit 42 its port, bits 43-44 the base of that step, bits 45-47 bases
        // inside the run, bits 48-61 those bases
        const u64 wfar = (!no_next && len - 1 <= KC_SC_NEXT_MAX) ? link[(base + far) * 2 + (fo == 0 ? 1u : 0u)] : 0;
        const bool far_here = DN ? LK_SLOT(wfar) - (set.rlo >> set.bshift) * (u64)set.nstride < (set.rcap >> set.bshift) * (u64)set.nstride : LK_SLOT(wfar) - set.rlo < set.rcap;
        if (!no_next && len - 1 <= KC_SC_NEXT_MAX && LK_IS_MATCHED(wfar) && far_here) {   // (a link that leaves the rank ends a fragment there: its far node must stay visible)
            link[(base + s) * 2 + xp] = LK_REJ | KC_SC_NEXT | (LK_SLOT(wfar) << 6) | (LK_PORT(wfar) << 42) | (LK_BASE(wfar) << 43) | ((u64)(len - 1) << 45) |
                                        ((bases >> 24) << 48);
        } else if (len - 1 <= KC_SC_INLINE_MAX) {
            link[(base + s) * 2 + xp] = LK_REJ | (far_field(far, s) << 6) | ((u64)fo << 20) | ((u64)(len - 1) << 21) | (bases << 26);
        } else {
            const u32 nw = (len - 1 + 31) / 32;
            const u32 at = atomicAdd(&n_long, 1u);
            const u32 rel = atomicAdd(&n_words, nw);
            // (more long runs than the list holds, or a run of 2^14 nodes: that end keeps its plain link)
            if (at < long_cap && len - 1 < (1u << 14)) { longs[2 * at] = 2 * s + (u32)o; longs[2 * at + 1] = rel; }
        }
    }
    __syncthreads();
    if (n_long == 0) return;
    const u64 pcap_blk = pool_cap / gridDim.x, pbase = lblk * pcap_blk;   // (absolute: KcFrag::shard_cap is 0)
    if (threadIdx.x == 0) blk_pool_base[lblk] = pbase;
    if (n_words > pcap_blk) return;   // no pool space: the long runs of this block keep their plain links
    const u32 nl = n_long < long_cap ? n_long : long_cap;
    for (u32 i = threadIdx.x; i < nl; i += blockDim.x) {
        const u32 pid = longs[2 * i], rel = longs[2 * i + 1];
        const u32 s = pid >> 1; const int o = (int)(pid & 1);
        u32 far, len; int fo; u64 bases;
        walk(s, o, far, fo, len, bases, pool + pbase + rel, true);
        const u32 xp = o == 0 ? 1u : 0u;
        link[(base + s) * 2 + xp] = LK_REJ | KC_SC_EXT | (far_field(far, s) << 6) | ((u64)fo << 20) | ((u64)(len - 1) << 21) | ((DN ? pbase + (u64)rel : (u64)rel) << 35);
    }
}

// puts the logged link words back (the hop walk met a chain beyond the limit, or the input has cycles: the classic flow
// takes over on the original links).  nstride != 0: links addressed by dense node ids (ids per block).
static __global__ void __launch_bounds__(256) kc_frag_restore_kernel(u64 *link, const u64 *__restrict__ log, const u64 *__restrict__ blk_log, int bshift, u64 blk0,
                                                                     u32 nstride = 0) {
    const u64 blk = blockIdx.x;
    const u64 e = blk_log[blk];
    const u32 n = (u32)(e & 0xffffu);
    const u64 lb = e >> 16;
    const u64 first = nstride ? (blk0 + blk) * (u64)nstride : (blk0 + blk) << bshift;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 w = log[lb + i];
        link[(first << 1) + ((w >> 46) & 0x3fffu)] = w & ~(0x3fffull << 46);
    }
}

// The k-mers NO walk reached, from SPARSE marks (long-chain inputs; dense node ids).  A segment walk that marked every node it
// passes wrote one scattered byte per node and direction -- three times the cost of the walk itself (measured on 2.4e9 genome
// k-mers: 335 ms against 116 ms).  The walks therefore mark only where a chain ENTERS a block (the node reached by a link from
// another block, a chain's head, its tail): one mark per in-block run and direction.  This kernel turns those marks back into the
// list of unvisited k-mers: one workgroup per block rebuilds the block's runs in LDS exactly as kc_frag_contract_kernel does (same
// digest, same ends), and a run whose entry end carries no mark -- no walk came in there -- is on a cycle, all of it; so are the
// nodes of cycles that never leave the block (they have no end at all).  FILL = false counts, FILL = true lists (node ids).
// log != nullptr: the links are CONTRACTED (kc_frag_contract_kernel); the block's logged words go back into the digest first, and a
// block that lists anything puts them back into the link array as well (FILL) and gives up its log -- the cycle rule reads and
// rewrites plain words there, the other blocks keep their shortcuts for the next pass.
template <int W, bool FILL>
__global__ void __launch_bounds__(1024) kc_frag_unvisited_kernel(SetView<W> set, u64 *link, int complements, const u8 *__restrict__ visited,
                                                                 u64 *out, u64 out_cap, u64 *cursor, const u64 *__restrict__ log = nullptr, u64 *blk_log = nullptr) {
    // LDS: the digest (one 32-bit word per port, see kc_frag_contract_kernel; bit 29 of a node's port-0 word: on a run that was
    // walked here, bit 28: marked), the ends (2 * bs u16), the unvisited nodes of the block (bs u16)
    extern __shared__ u32 kc_fl[];
    __shared__ u32 n_ent, n_out;
    __shared__ u64 g_base;
    const u32 bs = 1u << set.bshift;
    const u64 blk = (set.rlo >> set.bshift) + blockIdx.x;
    const u64 base = blk * (u64)set.nstride;
    const u32 nn = set.ncount[blk];
    unsigned short *ents = reinterpret_cast<unsigned short *>(kc_fl + 2 * (size_t)bs);
    unsigned short *lst = ents + 2 * (size_t)bs;
    if (threadIdx.x == 0) { n_ent = 0; n_out = 0; }
    for (u32 s = threadIdx.x; s < nn; s += blockDim.x) {
        const ulonglong2 ww = *reinterpret_cast<const ulonglong2 *>(link + (base + s) * 2);
        const u64 w2[2] = {ww.x, ww.y};
        const u32 vis = visited[base + s] ? 1u << 28 : 0u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const u64 w = w2[p];
            const bool in = LK_IS_MATCHED(w) && LK_SLOT(w) - base < (u64)nn;
            kc_fl[2 * s + p] = 0x40000000u | (p == 0 ? vis : 0u) | (in ? (0x80000000u | (u32)(LK_SLOT(w) - base) | ((u32)LK_PORT(w) << 14)) : 0u);
        }
    }
    u32 nlog = 0; u64 lb = 0;
    if (log) {
        const u64 e = blk_log[blockIdx.x];
        nlog = (u32)(e & 0xffffu); lb = e >> 16;
        __syncthreads();
        for (u32 i = threadIdx.x; i < nlog; i += blockDim.x) {   // (a logged word is an in-block link by construction)
            const u64 w = log[lb + i];
            const u32 lp = (u32)((w >> 46) & 0x3fffu);
            kc_fl[lp] = (kc_fl[lp] & (0x40000000u | (1u << 28))) | 0x80000000u | (u32)(LK_SLOT(w) - base) | ((u32)LK_PORT(w) << 14);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (u32 s0 = 0; s0 < nn; s0 += blockDim.x) {
        const u32 s = s0 + threadIdx.x;
        for (int o = 0; o < (complements ? 2 : 1); ++o) {
            const u32 f_in = s < nn ? kc_fl[2 * s + o] : 0u, f_out = s < nn ? kc_fl[2 * s + (o == 0 ? 1 : 0)] : 0u;
            const bool is_end = (f_in & 0x40000000u) && !(f_in >> 31) && (f_out >> 31);
            const unsigned long long m = __ballot(is_end);
            if (!m) continue;
            u32 at0 = 0;
            if (lane == 0) at0 = atomicAdd(&n_ent, (u32)__popcll(m));
            at0 = __shfl(at0, 0, 64);
            if (is_end) ents[at0 + (u32)__popcll(m & (((unsigned long long)1 << lane) - 1))] = (unsigned short)(2 * s + (u32)o);   // (at most 2 nn ends)
        }
    }
    __syncthreads();
    const u32 ne = n_ent;
    for (u32 e = threadIdx.x; e < ne; e += blockDim.x) {
        const u32 s = ents[e] >> 1;
        const int o = (int)(ents[e] & 1);
        // walk the run, flag its nodes; an unmarked run is listed by the walk from its smaller end (unidirectional: its only end)
        u32 cur = s, len = 1; int co = o;
        atomicOr(&kc_fl[2 * cur], 1u << 29);
        while (len <= bs) {
            const u32 w = kc_fl[2 * cur + (co == 0 ? 1 : 0)];
            if (!(w >> 31)) break;
            cur = w & 0x3fffu; co = (int)((w >> 14) & 1); ++len;
            atomicOr(&kc_fl[2 * cur], 1u << 29);
        }
        const bool marked = (kc_fl[2 * s] >> 28) & 1u;
        if (marked || (complements && cur < s)) continue;
        u32 at = atomicAdd(&n_out, len);
        cur = s; co = o;
        for (u32 i = 0; i < len; ++i) {
            if (at + i < bs) lst[at + i] = (unsigned short)cur;
            const u32 w = kc_fl[2 * cur + (co == 0 ? 1 : 0)];
            cur = w & 0x3fffu; co = (int)((w >> 14) & 1);
        }
    }
    __syncthreads();
    // nodes on no run: alone in the block (unvisited iff unmarked), or on a cycle inside the block
    for (u32 s = threadIdx.x; s < nn; s += blockDim.x) {
        const u32 f0 = kc_fl[2 * s], f1 = kc_fl[2 * s + 1];
        if ((f0 >> 29) & 1u) continue;
        const bool in0 = f0 >> 31, in1 = f1 >> 31;
        // (a node with exactly one link inside the block is the end of a run and carries the flag)
        const bool unv = (!in0 && !in1) ? !((f0 >> 28) & 1u) : (in0 && in1);
        if (!unv) continue;
        const u32 at = atomicAdd(&n_out, 1u);
        if (at < bs) lst[at] = (unsigned short)s;
    }
    __syncthreads();
    const u32 no = n_out < bs ? n_out : bs;
    if (no == 0) return;
    if (threadIdx.x == 0) g_base = atomicAdd((ull *)cursor, (ull)no);
    __syncthreads();
    if (FILL) {
        for (u32 i = threadIdx.x; i < no; i += blockDim.x)
            if (g_base + i < out_cap) out[g_base + i] = base + (u64)lst[i];   // (the cursor counts on: the caller sees an overflow)
        if (log && nlog) {
            for (u32 i = threadIdx.x; i < nlog; i += blockDim.x) {
                const u64 w = log[lb + i];
                link[(base << 1) + ((w >> 46) & 0x3fffu)] = w & ~(0x3fffull << 46);
            }
            if (threadIdx.x == 0) blk_log[blockIdx.x] = 0;
        }
    }
}

// measure pass: one thread per directed chain head. Marks visited nodes, decides which of the two directions of a
// chain is emitted (bi: the one starting at the smaller key) and records the winners.
// MARK: also flag visited nodes (only needed when the node count shows that cycles exist).
// REC: also record the bases appended along the walk (32 per word, first base in bits 63:62) into 32-word chunks drawn
// from `pool`, so that the winning direction is later emitted by copying its recording instead of walking the chain a
// third time (every step of a walk is a scattered 128-byte line fetch; the recording is 2 bits per step).
struct KcFrag {          // fragment shortcuts in the link words (kc_frag_contract_kernel); pool == nullptr: none
    const u64 *pool;
    const u64 *blk_pool_base;
    u64 shard_cap;
    u64 blk0;        // first block of the range the contraction ran over (hash-sharded tables: the rank's first block)
};
