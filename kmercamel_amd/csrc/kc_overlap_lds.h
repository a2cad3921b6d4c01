// LDS-resident replay of one key bucket (pair) of the greedy overlap matching -- same event order and decisions as
// kc_match_bucket (kc_overlap_logic.h, restating global.h:63-128), but every piece of mutable state lives in the LDS of
// the wavefront that owns the bucket.  The global-memory replay pays one L2 round trip per dependent access AND a drain
// of the previous event's stores (vmcnt counts stores on gfx950): ~4 us per event.  Here a wavefront loads the bucket
// (vertex ids, used flags, path ends) cooperatively, lane 0 replays the events out of LDS (~64-cycle accesses), and the
// wavefront writes the accepted edges back in parallel.
//
// State per suffix-side member A[i]: usedA[i] (A[i] has an out-edge == mirror(A[i]) has an in-edge), firstA[i] = first
// vertex of the path ending at A[i].  Per prefix-side member B[j]: usedB[j], lastB[j] = last vertex of the path starting
// at B[j].  The mirrored lists of PAIR mode share these entries (accessFirstLast's trick, global.h:28-29).  Every stored
// path end carries the indices under which it is itself a member of this bucket ("who has to be told when my path
// changes"), so an accepted edge updates at most four entries without any search:
//   path-first vertex F is stored as {F, iB(F): index of F in B, iA(mF): index of mirror(F) in A}
//   path-last  vertex L is stored as {L, iA(L): index of L in A, iB(mL): index of mirror(L) in B}
// Path ends that are not members are never read again by this bucket, so they are not tracked (other buckets never trust
// them either, see kc_overlap_logic.h); the exact ends are recomputed after the level by ov_fix_kernel.
#pragma once
#include "kc_overlap_logic.h"

#if defined(__HIPCC__)

#define KC_LDS_NONE 0xffffffffu
struct LdsEnd { u32 v, i1, i2; };   // vertex, index in its natural list, index of its mirror in the other list

struct LdsBucket {
    u64 *A, *B;            // vertex ids (ascending)
    LdsEnd *firstA, *lastB;
    u32 *piA, *piB;        // PAL only: index in B of mirror(A[i]) / index in A of mirror(B[j])
    u32 *evU, *evC;        // accepted events: (from<<31 | index), (candidate index)
    u32 *scr;              // unmatched events between parts: (from<<31 | index)
    u8 *usedA, *usedB, *evPart;
};

__device__ __forceinline__ size_t kc_lds_bytes(u64 nA, u64 nB) {
    u64 n = nA + nB;
    return (size_t)(n * 8 + n * sizeof(LdsEnd) + n * 4 + n * 4 * 3 + n * 3 + 64);
}
__host__ __device__ __forceinline__ size_t kc_lds_bytes_bound(u64 n) { return (size_t)(n * 8 + n * 12 + n * 4 + n * 12 + n * 3 + 64); }

__device__ __forceinline__ u32 lds_find(const u64 *a, u32 n, u64 v) {   // index of v in the ascending list a, or NONE
    u32 lo = 0, hi = n;
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return (lo < n && a[lo] == v) ? lo : KC_LDS_NONE;
}

// ascending order of the n vertex ids in a[] (LDS), by their rank among themselves (ids are distinct): every lane places its
// elements; tmp[] = n words of LDS.  n^2 / 64 compares per lane -- for the member lists of the grouped levels, which arrive unordered.
__device__ __forceinline__ void kc_lds_rank_sort(u64 *a, u32 n, u32 *tmp) {
    const int lane = threadIdx.x & 63;
    for (u32 i = lane; i < n; i += 64) tmp[i] = (u32)a[i];
    __syncthreads();
    for (u32 i = lane; i < n; i += 64) {
        const u32 v = tmp[i];
        u32 r = 0;
        for (u32 j = 0; j < n; ++j) r += tmp[j] < v ? 1u : 0u;
        a[r] = (u64)v;
    }
    __syncthreads();
}

// one wavefront (blockDim = 64); smem = dynamic LDS of at least kc_lds_bytes(nA, nB).  unsorted: gA / gB come in any order (the
// grouped levels scatter the members of a bucket through atomic cursors) and are put in ascending order here.
__device__ void kc_match_bucket_lds(const OvArgs &g, const u64 *gA, u32 nA, const u64 *gB, u32 nB, int mode, unsigned char *smem, bool unsorted = false) {
    const int lane = threadIdx.x & 63;
    const u64 S = g.S;
    const bool bi = g.complements != 0;
    const u32 n = nA + nB;
    LdsBucket b;
    unsigned char *p = smem;
    b.A = (u64 *)p; p += (size_t)nA * 8;
    b.B = (u64 *)p; p += (size_t)nB * 8;
    b.firstA = (LdsEnd *)p; p += (size_t)nA * sizeof(LdsEnd);
    b.lastB = (LdsEnd *)p; p += (size_t)nB * sizeof(LdsEnd);
    b.piA = (u32 *)p; p += (size_t)nA * 4;
    b.piB = (u32 *)p; p += (size_t)nB * 4;
    b.evU = (u32 *)p; p += (size_t)n * 4;
    b.evC = (u32 *)p; p += (size_t)n * 4;
    b.scr = (u32 *)p; p += (size_t)n * 4;
    b.usedA = (u8 *)p; p += nA;
    b.usedB = (u8 *)p; p += nB;
    b.evPart = (u8 *)p;

    // ---- cooperative load
    for (u32 i = lane; i < nA; i += 64) { b.A[i] = gA[i]; b.usedA[i] = 0; }
    for (u32 j = lane; j < nB; j += 64) { b.B[j] = gB[j]; b.usedB[j] = 0; }
    __syncthreads();
    if (unsorted) { kc_lds_rank_sort(b.A, nA, b.evU); kc_lds_rank_sort(b.B, nB, b.evU); }   // (evU: n words, free until the replay)
    for (u32 i = lane; i < nA; i += 64) {
        u64 F = g.lower_bound ? 0 : g.first_snap[b.A[i]];
        LdsEnd e;
        e.v = (u32)F;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.B, nB, F);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.A, nA, ov_mirror(F, S)) : KC_LDS_NONE;
        b.firstA[i] = e;
        b.piA[i] = (mode == KC_MODE_PAL) ? lds_find(b.B, nB, ov_mirror(b.A[i], S)) : KC_LDS_NONE;
    }
    for (u32 j = lane; j < nB; j += 64) {
        u64 L = g.lower_bound ? 0 : g.last_snap[b.B[j]];
        LdsEnd e;
        e.v = (u32)L;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.A, nA, L);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.B, nB, ov_mirror(L, S)) : KC_LDS_NONE;
        b.lastB[j] = e;
        b.piB[j] = (mode == KC_MODE_PAL) ? lds_find(b.A, nA, ov_mirror(b.B[j], S)) : KC_LDS_NONE;
    }
    __syncthreads();

    // Live events are kept in two linked lists in reference order (list 1: the members of A ascending; list 2, PAIR mode
    // only: the members of B viewed through mirror(), ascending).  b.scr[i] = next live event of list 1 after A[i],
    // b.scr[nA + j] = next of list 2 after B[j].  A matched (or otherwise consumed) event is unlinked, so later parts never
    // see it again, and a list whose candidates are exhausted in the current part is not traversed at all.
    const u32 splitB = (mode == KC_MODE_PAIR) ? (u32)ov_split(b.B, nB, S) : 0;
    for (u32 i = lane; i < nA; i += 64) b.scr[i] = (i + 1 < nA) ? i + 1 : KC_LDS_NONE;
    for (u32 j = lane; j < nB; j += 64) {
        u32 nx = (j + 1 < nB) ? j + 1 : KC_LDS_NONE;
        if (splitB > 0 && splitB < nB) {          // order: splitB .. nB-1, then 0 .. splitB-1
            if (j == nB - 1) nx = 0;
            if (j == splitB - 1) nx = KC_LDS_NONE;
        }
        b.scr[nA + j] = nx;
    }
    __syncthreads();

    __shared__ u32 nev_shared;
    if (lane == 0) {
        // ---- sequential replay (reference event order: part, then suffix vertex ascending; candidates descending)
        OvList C1{b.B, nB, 0, S, 0}, C2{b.A, 0, 0, S, 1};
        if (mode == KC_MODE_PAIR) { C2.n = nA; C2.split = ov_split(b.A, nA, S); }
        // index of the t-th element of a mirrored list inside its base array
        auto mirrored_index = [](const OvList &l, u64 t) -> u32 { u64 hi = l.n - l.split; return (u32)(t < hi ? l.split + t : t - hi); };
        u32 head1 = nA ? 0 : KC_LDS_NONE;
        u32 head2 = (mode == KC_MODE_PAIR && nB) ? ((splitB < nB) ? splitB : 0) : KC_LDS_NONE;
        u32 nev = 0;
        u64 left = (u64)nA + (mode == KC_MODE_PAIR ? (u64)nB : 0);
        u64 nx1 = 0, nx2 = 0;   // first candidate of C1 / C2 not yet assigned to a part (empty parts are skipped, cf. kc_match_bucket)
        while (left > 0 && (head1 != KC_LDS_NONE || head2 != KC_LDS_NONE)) {
            const u64 v1 = nx1 < C1.n ? C1.get(nx1) : ~(u64)0, v2 = nx2 < C2.n ? C2.get(nx2) : ~(u64)0;
            const u64 vmin = v1 < v2 ? v1 : v2;
            if (vmin == ~(u64)0) break;
            const u64 part = vmin / g.batch, hi_v = (part + 1) * g.batch;
            const u64 cl1 = nx1, cl2 = nx2;
            while (nx1 < C1.n && C1.get(nx1) < hi_v) ++nx1;
            while (nx2 < C2.n && C2.get(nx2) < hi_v) ++nx2;
            u64 top1 = nx1, top2 = nx2;
            u32 cur1 = head1, prev1 = KC_LDS_NONE, cur2 = head2, prev2 = KC_LDS_NONE;
            u64 u1 = cur1 != KC_LDS_NONE ? b.A[cur1] : 0, u2 = cur2 != KC_LDS_NONE ? ov_mirror(b.B[cur2], S) : 0;
            while (true) {
                while (top1 > cl1 && b.usedB[top1 - 1]) --top1;                              // used candidates never come back
                while (top2 > cl2 && b.usedA[mirrored_index(C2, top2 - 1)]) --top2;
                const bool can1 = cur1 != KC_LDS_NONE && top1 > cl1, can2 = cur2 != KC_LDS_NONE && top2 > cl2;
                if (!can1 && !can2) break;                                                   // no event of this part can match any more
                const int from = (can1 && (!can2 || u1 <= u2)) ? 1 : 2;                      // event: from 1 = A[eidx], from 2 = mirror(B[eidx])
                const u32 eidx = (from == 1) ? cur1 : cur2;
                const u64 u = (from == 1) ? u1 : u2;
                const u32 nxt = b.scr[(from == 1 ? 0 : nA) + eidx];
                const bool used = (from == 1) ? b.usedA[eidx] : b.usedB[eidx];               // suffixForbidden (global.h:87)
                bool matched = false;
                if (!used) {
                    // first vertex of u's path, as a path-first triple
                    LdsEnd F;
                    if (from == 1) F = b.firstA[eidx];
                    else { LdsEnd L = b.lastB[eidx]; F.v = (u32)ov_mirror(L.v, S); F.i1 = L.i2; F.i2 = L.i1; }
                    const u64 top = (from == 1) ? top1 : top2;
                    const u64 cl = (from == 1) ? cl1 : cl2;
                    auto cand_index = [&](u64 t) -> u32 { return (from == 1) ? (u32)t : mirrored_index(C2, t); };
                    auto cand_used = [&](u32 ci) -> bool { return (from == 1) ? b.usedB[ci] : b.usedA[ci]; };
                    for (u64 t = top; t > cl; --t) {
                        const u32 ci = cand_index(t - 1);
                        if (cand_used(ci)) continue;                                 // prefixForbidden (global.h:97)
                        const u64 c = (from == 1) ? b.B[ci] : ov_mirror(b.A[ci], S);
                        if (!g.lower_bound) {
                            if (bi && c == ov_mirror(u, S)) continue;                // complementary (global.h:93)
                            if ((u64)F.v == c) continue;                             // would close a cycle (global.h:95)
                            bool banned = false;
                            for (u64 q = 0; q < g.nbl; ++q) banned = banned || (g.bl[2 * q] == u && g.bl[2 * q + 1] == c);
                            if (banned) continue;
                        }
                        // ---- accept u -> c (+ mirror(c) -> mirror(u))
                        if (from == 1) { b.usedA[eidx] = 1; b.usedB[ci] = 1; } else { b.usedB[eidx] = 1; b.usedA[ci] = 1; }
                        if (mode == KC_MODE_PAL) {       // the mirrored edge consumes members of the same lists
                            u32 ma = b.piB[ci], mb = b.piA[eidx];     // mirror(c) in A, mirror(u) in B
                            if (ma != KC_LDS_NONE) b.usedA[ma] = 1;
                            if (mb != KC_LDS_NONE) b.usedB[mb] = 1;
                        }
                        if (!g.lower_bound) {
                            LdsEnd L;                    // last vertex of c's path, as a path-last triple
                            if (from == 1) L = b.lastB[ci];
                            else { LdsEnd Fc = b.firstA[ci]; L.v = (u32)ov_mirror(Fc.v, S); L.i1 = Fc.i2; L.i2 = Fc.i1; }
                            // new path F .. u c .. L (and its mirror): tell the members among the four end vertices
                            if (L.i1 != KC_LDS_NONE) b.firstA[L.i1] = F;                                       // first_of(L) = F
                            if (L.i2 != KC_LDS_NONE) { LdsEnd m; m.v = (u32)ov_mirror(F.v, S); m.i1 = F.i2; m.i2 = F.i1; b.lastB[L.i2] = m; }   // last_of(mL) = mF
                            if (F.i1 != KC_LDS_NONE) b.lastB[F.i1] = L;                                        // last_of(F) = L
                            if (F.i2 != KC_LDS_NONE) { LdsEnd m; m.v = (u32)ov_mirror(L.v, S); m.i1 = L.i2; m.i2 = L.i1; b.firstA[F.i2] = m; }  // first_of(mF) = mL
                        }
                        b.evU[nev] = ((u32)(from == 2) << 31) | eidx;
                        b.evC[nev] = ci;
                        b.evPart[nev] = (u8)part;
                        ++nev;
                        left -= 1;
                        if (bi && (!g.lower_bound || ov_mirror(c, S) != u)) left -= (left > 0);
                        matched = true;
                        break;
                    }
                }
                // advance; a used or matched event leaves its list for good
                if (from == 1) {
                    if (used || matched) { if (prev1 == KC_LDS_NONE) head1 = nxt; else b.scr[prev1] = nxt; } else prev1 = eidx;
                    cur1 = nxt;
                    if (cur1 != KC_LDS_NONE) u1 = b.A[cur1];
                } else {
                    if (used || matched) { if (prev2 == KC_LDS_NONE) head2 = nxt; else b.scr[nA + prev2] = nxt; } else prev2 = eidx;
                    cur2 = nxt;
                    if (cur2 != KC_LDS_NONE) u2 = ov_mirror(b.B[cur2], S);
                }
            }
        }
        nev_shared = nev;
    }
    __syncthreads();

    // ---- parallel write-back of the accepted events (edge arrays + new-edge log)
    const u32 nev = nev_shared;
    if (nev == 0) return;
    __shared__ u64 base_shared;
    __shared__ u64 *log_shared;
    if (lane == 0) {
        const u64 want = (u64)nev * (bi ? 2 : 1);
        if (g.ne_sh) {   // sharded log (see ov_add_event)
            const u32 sh = ((u32)blockIdx.x * 0x9E3779B1u) >> 26;   // (one bucket per workgroup)
            const u64 pos = atomicAdd((unsigned long long *)&g.ne_shcur[sh * 16], (unsigned long long)want);
            const bool fits = pos + want <= g.ne_shcap;
            if (!fits) g.ne_shcur[64 * 16] = 1;
            log_shared = fits ? g.ne_sh + ((u64)sh * g.ne_shcap + pos) * 5 : nullptr;
            base_shared = 0;
        } else {
            base_shared = atomicAdd((unsigned long long *)g.ne_count, (unsigned long long)want);
            log_shared = g.ne;
        }
    }
    __syncthreads();
    const u64 base = base_shared;
    u64 *const logp = log_shared;
    for (u32 e = lane; e < nev; e += 64) {
        const u32 ev = b.evU[e], eidx = ev & 0x7fffffffu, ci = b.evC[e];
        const bool from2 = (ev >> 31) != 0;
        const u64 u = from2 ? ov_mirror(b.B[eidx], S) : b.A[eidx];
        const u64 c = from2 ? ov_mirror(b.A[ci], S) : b.B[ci];
        const u64 part = b.evPart[e];
        g.edge_from[u] = c; g.overlaps[u] = (u8)g.d; g.pred[c] = u;
        u64 *le = logp ? logp + (base + (u64)e * (bi ? 2 : 1)) * 5 : nullptr;
        if (le) { le[0] = u; le[1] = c; le[2] = u; le[3] = c; le[4] = part; }
        if (bi) {
            const u64 x2 = ov_mirror(c, S), y2 = ov_mirror(u, S);
            g.edge_from[x2] = y2; g.overlaps[x2] = (u8)g.d; g.pred[y2] = x2;
            if (le) { le[5] = x2; le[6] = y2; le[7] = u; le[8] = c; le[9] = part; }
        }
    }
}

#endif  // __HIPCC__
