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

#define KC_LDS_NONE 0xffffffffu
struct LdsEnd { u32 v, i1, i2; };   // vertex, index in its natural list, index of its mirror in the other list

// (vertex ids are kept as 32-bit words in LDS: a bucket's replay is a chain of dependent LDS accesses and integer instructions issued
// by ONE lane -- ~1.7 us per event with 64-bit ids and the three bucket kinds in one body, measured on the single bucket of d = 0 --
// so the replay below is compiled per kind, with the list of mirrored events only where it exists.  2 S < 2^32 vertices: LdsEnd.v.)
struct LdsBucket {
    u32 *A, *B;            // vertex ids (ascending)
    LdsEnd *firstA, *lastB;
    u32 *piA, *piB;        // PAL only: index in B of mirror(A[i]) / index in A of mirror(B[j])
    u32 *evU, *evC;        // accepted events: (from<<31 | index), (candidate index)
    u32 *scr;              // next live event of list 1 after A[i] / of list 2 after B[j] (at nA + j)
    u8 *usedA, *usedB, *evPart;
};

KC_HD size_t kc_lds_bytes(u64 nA, u64 nB) {
    u64 n = nA + nB;
    return (size_t)(n * 8 + n * sizeof(LdsEnd) + n * 4 + n * 4 * 3 + n * 3 + 64);
}
KC_HD size_t kc_lds_bytes_bound(u64 n) { return (size_t)(n * 8 + n * 12 + n * 4 + n * 12 + n * 3 + 64); }

KC_HD u32 lds_find(const u32 *a, u32 n, u32 v) {   // index of v in the ascending list a, or NONE
    u32 lo = 0, hi = n;
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return (lo < n && a[lo] == v) ? lo : KC_LDS_NONE;
}
KC_HD u32 lds_lower(const u32 *a, u32 lo, u32 n, u32 v) {   // first index >= lo with a[idx] >= v
    u32 hi = n;
    while (lo < hi) { u32 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
KC_HD u32 lds_mirror(u32 v, u32 S) { return v < S ? v + S : v - S; }

#if defined(__HIPCC__)
// ascending order of the n vertex ids in a[] (LDS), by their rank among themselves (ids are distinct), for the member lists of the
// grouped levels, which arrive unordered; tmp[] = n words of LDS.  Every lane keeps Q of the elements in registers and runs ONCE over the
// list for them (four broadcast reads in flight): n^2 / 64 compares per lane, two instructions each.  (One element per lane and one
// dependent LDS read per compare took 1.1 ms for each list of the single bucket of d = 0 -- 1 326 members -- at cfg[2]: most of that
// level's 2.6 ms, and of every level below d = 5.)
template <int Q>
__device__ __forceinline__ void kc_lds_rank_sort_q(u32 *a, u32 n, const u32 *tmp) {
    const u32 lane = threadIdx.x & 63;
    for (u32 base = 0; base < n; base += 64 * Q) {
        u32 v[Q], r[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { const u32 i = base + (u32)q * 64 + lane; v[q] = i < n ? tmp[i] : 0u; r[q] = 0; }
        u32 j = 0;
        for (; j + 4 <= n; j += 4) {
            const u32 x0 = tmp[j], x1 = tmp[j + 1], x2 = tmp[j + 2], x3 = tmp[j + 3];
#pragma unroll
            for (int q = 0; q < Q; ++q) r[q] += (x0 < v[q] ? 1u : 0u) + (x1 < v[q] ? 1u : 0u) + (x2 < v[q] ? 1u : 0u) + (x3 < v[q] ? 1u : 0u);
        }
        for (; j < n; ++j) {
            const u32 x = tmp[j];
#pragma unroll
            for (int q = 0; q < Q; ++q) r[q] += x < v[q] ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) { const u32 i = base + (u32)q * 64 + lane; if (i < n) a[r[q]] = v[q]; }
    }
}
__device__ __forceinline__ void kc_lds_rank_sort(u32 *a, u32 n, u32 *tmp) {
    const int lane = threadIdx.x & 63;
    for (u32 i = lane; i < n; i += 64) tmp[i] = a[i];
    __syncthreads();
    if (n <= 128) kc_lds_rank_sort_q<2>(a, n, tmp); else kc_lds_rank_sort_q<8>(a, n, tmp);
    __syncthreads();
}

#endif  // __HIPCC__

// The sequential replay of one bucket (reference event order: part, then suffix vertex ascending; candidates descending), run by ONE lane
// out of LDS; returns the number of accepted events (listed in evU / evC / evPart).  Live events are kept in two linked lists in
// reference order (list 1: the members of A ascending; list 2, PAIR only: the members of B viewed through mirror(), ascending): a matched
// (or otherwise consumed) event is unlinked, so later parts never see it again, and a list whose candidates are exhausted in the
// current part is not traversed at all.
template <int MODE, bool LB>
KC_HD u32 kc_lds_replay(const OvArgs &g, const LdsBucket &b, const u32 nA, const u32 nB, const u32 splitA, const u32 splitB) {
    constexpr bool BI = MODE != KC_MODE_UNI, PAIR = MODE == KC_MODE_PAIR;
    const u32 S = (u32)g.S, batch = (u32)g.batch;
    const u32 n2 = PAIR ? nA : 0, hi2 = n2 - splitA;   // the candidates of list 2: A viewed through mirror(), ascending: element t sits at ...
    auto idx2 = [&](u32 t) -> u32 { return t < hi2 ? splitA + t : t - hi2; };
    u32 head1 = nA ? 0 : KC_LDS_NONE;
    u32 head2 = (PAIR && nB) ? ((splitB < nB) ? splitB : 0) : KC_LDS_NONE;
    u32 nev = 0;
    u32 left = nA + (PAIR ? nB : 0);
    u32 nx1 = 0, nx2 = 0;   // first candidate of C1 / C2 not yet assigned to a part (empty parts are skipped, cf. kc_match_bucket)
    while (left > 0 && (head1 != KC_LDS_NONE || head2 != KC_LDS_NONE)) {
        const u32 v1 = nx1 < nB ? b.B[nx1] : 0xffffffffu;
        const u32 v2 = (PAIR && nx2 < n2) ? lds_mirror(b.A[idx2(nx2)], S) : 0xffffffffu;
        const u32 vmin = v1 < v2 ? v1 : v2;
        if (vmin == 0xffffffffu) break;
        const u32 part = vmin / batch;
        const u64 hi64 = (u64)(part + 1) * batch;
        const u32 hi_v = hi64 > 0xffffffffull ? 0xffffffffu : (u32)hi64;
        const u32 cl1 = nx1, cl2 = nx2;
        nx1 = lds_lower(b.B, nx1, nB, hi_v);
        if (PAIR) {   // (the mirrored view: its first hi2 elements are A[splitA ..) - S, the others A[0 .. splitA) + S)
            if (nx2 < hi2) {
                const u32 e = hi_v >= 0xffffffffu - S ? nA : lds_lower(b.A, splitA + nx2, nA, hi_v + S);
                nx2 = e - splitA;
            }
            if (nx2 >= hi2) nx2 = hi2 + (hi_v > S ? lds_lower(b.A, nx2 - hi2, splitA, hi_v - S) : (nx2 - hi2));
        }
        u32 top1 = nx1, top2 = nx2;
        u32 cur1 = head1, prev1 = KC_LDS_NONE, cur2 = head2, prev2 = KC_LDS_NONE;
        u32 u1 = cur1 != KC_LDS_NONE ? b.A[cur1] : 0, u2 = (PAIR && cur2 != KC_LDS_NONE) ? lds_mirror(b.B[cur2], S) : 0;
        while (true) {
            while (top1 > cl1 && b.usedB[top1 - 1]) --top1;                              // used candidates never come back
            if (PAIR) while (top2 > cl2 && b.usedA[idx2(top2 - 1)]) --top2;
            const bool can1 = cur1 != KC_LDS_NONE && top1 > cl1, can2 = PAIR && cur2 != KC_LDS_NONE && top2 > cl2;
            if (!can1 && !can2) break;                                                   // no event of this part can match any more
            const bool f2 = PAIR && !(can1 && (!can2 || u1 <= u2));                      // event: from list 1 = A[eidx], from list 2 = mirror(B[eidx])
            bool matched = false, used;
            u32 nxt;
            if (!f2) {
                const u32 eidx = cur1, u = u1;
                nxt = b.scr[eidx];
                used = b.usedA[eidx] != 0;                                               // suffixForbidden (global.h:87)
                if (!used) {
                    const LdsEnd F = b.firstA[eidx];                                     // first vertex of u's path, as a path-first triple
                    for (u32 t = top1; t > cl1; --t) {
                        const u32 ci = t - 1;
                        if (b.usedB[ci]) continue;                                       // prefixForbidden (global.h:97)
                        const u32 c = b.B[ci];
                        if (!LB) {
                            if (BI && c == lds_mirror(u, S)) continue;                   // complementary (global.h:93)
                            if (F.v == c) continue;                                      // would close a cycle (global.h:95)
                            bool banned = false;
                            for (u64 q = 0; q < g.nbl; ++q) banned = banned || (g.bl[2 * q] == (u64)u && g.bl[2 * q + 1] == (u64)c);
                            if (banned) continue;
                        }
                        // ---- accept u -> c (+ mirror(c) -> mirror(u))
                        b.usedA[eidx] = 1; b.usedB[ci] = 1;
                        if (MODE == KC_MODE_PAL) {       // the mirrored edge consumes members of the same lists
                            const u32 ma = b.piB[ci], mb = b.piA[eidx];     // mirror(c) in A, mirror(u) in B
                            if (ma != KC_LDS_NONE) b.usedA[ma] = 1;
                            if (mb != KC_LDS_NONE) b.usedB[mb] = 1;
                        }
                        if (!LB) {
                            const LdsEnd L = b.lastB[ci];   // last vertex of c's path, as a path-last triple
                            // new path F .. u c .. L (and its mirror): tell the members among the four end vertices
                            if (L.i1 != KC_LDS_NONE) b.firstA[L.i1] = F;                                                          // first_of(L) = F
                            if (BI && L.i2 != KC_LDS_NONE) { LdsEnd m; m.v = lds_mirror(F.v, S); m.i1 = F.i2; m.i2 = F.i1; b.lastB[L.i2] = m; }   // last_of(mL) = mF
                            if (F.i1 != KC_LDS_NONE) b.lastB[F.i1] = L;                                                           // last_of(F) = L
                            if (BI && F.i2 != KC_LDS_NONE) { LdsEnd m; m.v = lds_mirror(L.v, S); m.i1 = L.i2; m.i2 = L.i1; b.firstA[F.i2] = m; }  // first_of(mF) = mL
                        }
                        b.evU[nev] = eidx; b.evC[nev] = ci; b.evPart[nev] = (u8)part;
                        ++nev;
                        left -= 1;
                        if (BI && (!LB || lds_mirror(c, S) != u)) left -= (left > 0);
                        matched = true;
                        break;
                    }
                }
                // advance; a used or matched event leaves its list for good
                if (used || matched) { if (prev1 == KC_LDS_NONE) head1 = nxt; else b.scr[prev1] = nxt; } else prev1 = eidx;
                cur1 = nxt;
                if (cur1 != KC_LDS_NONE) u1 = b.A[cur1];
            } else {   // (PAIR only) the event mirror(B[eidx]); its candidates are the mirrors of A's members
                const u32 eidx = cur2, u = u2;
                nxt = b.scr[nA + eidx];
                used = b.usedB[eidx] != 0;
                if (!used) {
                    LdsEnd F;
                    { const LdsEnd L = b.lastB[eidx]; F.v = lds_mirror(L.v, S); F.i1 = L.i2; F.i2 = L.i1; }
                    for (u32 t = top2; t > cl2; --t) {
                        const u32 ci = idx2(t - 1);
                        if (b.usedA[ci]) continue;
                        const u32 c = lds_mirror(b.A[ci], S);
                        if (!LB) {
                            if (c == lds_mirror(u, S)) continue;
                            if (F.v == c) continue;
                            bool banned = false;
                            for (u64 q = 0; q < g.nbl; ++q) banned = banned || (g.bl[2 * q] == (u64)u && g.bl[2 * q + 1] == (u64)c);
                            if (banned) continue;
                        }
                        b.usedB[eidx] = 1; b.usedA[ci] = 1;
                        if (!LB) {
                            LdsEnd L;
                            { const LdsEnd Fc = b.firstA[ci]; L.v = lds_mirror(Fc.v, S); L.i1 = Fc.i2; L.i2 = Fc.i1; }
                            if (L.i1 != KC_LDS_NONE) b.firstA[L.i1] = F;
                            if (L.i2 != KC_LDS_NONE) { LdsEnd m; m.v = lds_mirror(F.v, S); m.i1 = F.i2; m.i2 = F.i1; b.lastB[L.i2] = m; }
                            if (F.i1 != KC_LDS_NONE) b.lastB[F.i1] = L;
                            if (F.i2 != KC_LDS_NONE) { LdsEnd m; m.v = lds_mirror(L.v, S); m.i1 = L.i2; m.i2 = L.i1; b.firstA[F.i2] = m; }
                        }
                        b.evU[nev] = 0x80000000u | eidx; b.evC[nev] = ci; b.evPart[nev] = (u8)part;
                        ++nev;
                        left -= 1;
                        if (!LB || lds_mirror(c, S) != u) left -= (left > 0);
                        matched = true;
                        break;
                    }
                }
                if (used || matched) { if (prev2 == KC_LDS_NONE) head2 = nxt; else b.scr[nA + prev2] = nxt; } else prev2 = eidx;
                cur2 = nxt;
                if (cur2 != KC_LDS_NONE) u2 = lds_mirror(b.B[cur2], S);
            }
        }
    }
    return nev;
}

#if defined(__HIPCC__)
// one wavefront (blockDim = 64); smem = dynamic LDS of at least kc_lds_bytes(nA, nB).  unsorted: gA / gB come in any order (the
// grouped levels scatter the members of a bucket through atomic cursors) and are put in ascending order here.
__device__ void kc_match_bucket_lds(const OvArgs &g, const u64 *gA, u32 nA, const u64 *gB, u32 nB, int mode, unsigned char *smem, bool unsorted = false) {
    const int lane = threadIdx.x & 63;
    const u32 S = (u32)g.S;
    const bool bi = g.complements != 0;
    const u32 n = nA + nB;
    LdsBucket b;
    unsigned char *p = smem;
    b.A = (u32 *)p; p += (size_t)nA * 4;
    b.B = (u32 *)p; p += (size_t)nB * 4;
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
    for (u32 i = lane; i < nA; i += 64) { b.A[i] = (u32)gA[i]; b.usedA[i] = 0; }
    for (u32 j = lane; j < nB; j += 64) { b.B[j] = (u32)gB[j]; b.usedB[j] = 0; }
    __syncthreads();
    if (unsorted) { kc_lds_rank_sort(b.A, nA, b.evU); kc_lds_rank_sort(b.B, nB, b.evU); }   // (evU: n words, free until the replay)
    for (u32 i = lane; i < nA; i += 64) {
        const u32 F = g.lower_bound ? 0 : (u32)g.first_snap[b.A[i]];
        LdsEnd e;
        e.v = F;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.B, nB, F);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.A, nA, lds_mirror(F, S)) : KC_LDS_NONE;
        b.firstA[i] = e;
        b.piA[i] = (mode == KC_MODE_PAL) ? lds_find(b.B, nB, lds_mirror(b.A[i], S)) : KC_LDS_NONE;
    }
    for (u32 j = lane; j < nB; j += 64) {
        const u32 L = g.lower_bound ? 0 : (u32)g.last_snap[b.B[j]];
        LdsEnd e;
        e.v = L;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.A, nA, L);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.B, nB, lds_mirror(L, S)) : KC_LDS_NONE;
        b.lastB[j] = e;
        b.piB[j] = (mode == KC_MODE_PAL) ? lds_find(b.A, nA, lds_mirror(b.B[j], S)) : KC_LDS_NONE;
    }
    // b.scr[i] = next live event of list 1 after A[i], b.scr[nA + j] = next of list 2 after B[j] (order: splitB .. nB-1, then 0 .. splitB-1)
    const u32 splitA = (mode == KC_MODE_PAIR) ? lds_lower(b.A, 0, nA, S) : 0;
    const u32 splitB = (mode == KC_MODE_PAIR) ? lds_lower(b.B, 0, nB, S) : 0;
    for (u32 i = lane; i < nA; i += 64) b.scr[i] = (i + 1 < nA) ? i + 1 : KC_LDS_NONE;
    for (u32 j = lane; j < nB; j += 64) {
        u32 nx = (j + 1 < nB) ? j + 1 : KC_LDS_NONE;
        if (splitB > 0 && splitB < nB) {
            if (j == nB - 1) nx = 0;
            if (j == splitB - 1) nx = KC_LDS_NONE;
        }
        b.scr[nA + j] = nx;
    }
    __syncthreads();

    __shared__ u32 nev_shared;
    if (lane == 0) {
        u32 nev;
        if (g.lower_bound) {
            nev = mode == KC_MODE_PAIR ? kc_lds_replay<KC_MODE_PAIR, true>(g, b, nA, nB, splitA, splitB)
                : mode == KC_MODE_PAL ? kc_lds_replay<KC_MODE_PAL, true>(g, b, nA, nB, 0, 0) : kc_lds_replay<KC_MODE_UNI, true>(g, b, nA, nB, 0, 0);
        } else {
            nev = mode == KC_MODE_PAIR ? kc_lds_replay<KC_MODE_PAIR, false>(g, b, nA, nB, splitA, splitB)
                : mode == KC_MODE_PAL ? kc_lds_replay<KC_MODE_PAL, false>(g, b, nA, nB, 0, 0) : kc_lds_replay<KC_MODE_UNI, false>(g, b, nA, nB, 0, 0);
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
        const u64 u = from2 ? lds_mirror(b.B[eidx], S) : b.A[eidx];
        const u64 c = from2 ? lds_mirror(b.A[ci], S) : b.B[ci];
        const u64 part = b.evPart[e];
        g.edge_from[u] = c; g.overlaps[u] = (u8)g.d; g.pred[c] = u;
        u64 *le = logp ? logp + (base + (u64)e * (bi ? 2 : 1)) * 5 : nullptr;
        if (le) { le[0] = u; le[1] = c; le[2] = u; le[3] = c; le[4] = part; }
        if (bi) {
            const u64 x2 = ov_mirror(c, g.S), y2 = ov_mirror(u, g.S);
            g.edge_from[x2] = y2; g.overlaps[x2] = (u8)g.d; g.pred[y2] = x2;
            if (le) { le[5] = x2; le[6] = y2; le[7] = u; le[8] = c; le[9] = part; }
        }
    }
}


#endif  // __HIPCC__

// ---- small buckets: ONE LANE per bucket, out of that lane's own stretch of LDS.  The thread-per-bucket replay from global memory
// (kc_match_bucket) pays ~7 dependent memory round trips per event, and the lanes of a wavefront wait for each other's: 0.4 - 1.1 ms per
// level for the 3 - 5 10^5 two-to-four-member buckets of the mid levels at cfg[2].  Here a lane loads its bucket's ids and path ends with
// all loads in flight at once, replays out of LDS (kc_lds_replay, the same code as a wavefront's bucket) and writes the edges back.
#define KC_LANE_MAX 11          /* members; larger buckets that reach the thread-per-bucket kernels keep the global replay */
#define KC_LANE_STRIDE 380      /* bytes per lane: 34 per member; 95 words -- odd, so lanes at the same offset use different banks */
KC_HD void kc_match_bucket_lane(const OvArgs &g, const u64 *gA, u32 nA, const u64 *gB, u32 nB, int mode, unsigned char *smem, bool unsorted, u32 tag) {
    const u32 S = (u32)g.S;
    const bool bi = g.complements != 0;
    const u32 n = nA + nB;
    LdsBucket b;
    unsigned char *p = smem;
    b.A = (u32 *)p; p += (size_t)nA * 4;
    b.B = (u32 *)p; p += (size_t)nB * 4;            // (B follows A, lastB follows firstA, usedB follows usedA: the loads below run over both)
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
#if defined(__HIP_DEVICE_COMPILE__)
    {
        u32 id[KC_LANE_MAX];
#pragma unroll
        for (int q = 0; q < KC_LANE_MAX; ++q) id[q] = (u32)q < n ? (u32)((u32)q < nA ? gA[q] : gB[(u32)q - nA]) : 0u;
#pragma unroll
        for (int q = 0; q < KC_LANE_MAX; ++q) if ((u32)q < n) { b.A[q] = id[q]; b.usedA[q] = 0; }
    }
#else   // (the host build -- tests/sim -- runs the same replay on buckets of any size out of ordinary memory)
    for (u32 q = 0; q < n; ++q) { b.A[q] = (u32)(q < nA ? gA[q] : gB[q - nA]); b.usedA[q] = 0; }
#endif
    if (unsorted) {
        for (u32 h = 0; h < 2; ++h) {
            u32 *a = h ? b.B : b.A;
            const u32 m = h ? nB : nA;
            for (u32 i = 1; i < m; ++i) {
                const u32 v = a[i];
                u32 j = i;
                while (j > 0 && a[j - 1] > v) { a[j] = a[j - 1]; --j; }
                a[j] = v;
            }
        }
    }
    if (!g.lower_bound) {
#if defined(__HIP_DEVICE_COMPILE__)
        u32 sn[KC_LANE_MAX];
#pragma unroll
        for (int q = 0; q < KC_LANE_MAX; ++q) sn[q] = (u32)q < n ? (u32)((u32)q < nA ? g.first_snap[b.A[q]] : g.last_snap[b.A[q]]) : 0u;
#pragma unroll
        for (int q = 0; q < KC_LANE_MAX; ++q) if ((u32)q < n) b.firstA[q].v = sn[q];
#else
        for (u32 q = 0; q < n; ++q) b.firstA[q].v = (u32)(q < nA ? g.first_snap[b.A[q]] : g.last_snap[b.A[q]]);
#endif
    }
    for (u32 i = 0; i < nA; ++i) {
        const u32 F = g.lower_bound ? 0 : b.firstA[i].v;
        LdsEnd e;
        e.v = F;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.B, nB, F);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.A, nA, lds_mirror(F, S)) : KC_LDS_NONE;
        b.firstA[i] = e;
        b.piA[i] = (mode == KC_MODE_PAL) ? lds_find(b.B, nB, lds_mirror(b.A[i], S)) : KC_LDS_NONE;
    }
    for (u32 j = 0; j < nB; ++j) {
        const u32 L = g.lower_bound ? 0 : b.lastB[j].v;
        LdsEnd e;
        e.v = L;
        e.i1 = g.lower_bound ? KC_LDS_NONE : lds_find(b.A, nA, L);
        e.i2 = (bi && !g.lower_bound) ? lds_find(b.B, nB, lds_mirror(L, S)) : KC_LDS_NONE;
        b.lastB[j] = e;
        b.piB[j] = (mode == KC_MODE_PAL) ? lds_find(b.A, nA, lds_mirror(b.B[j], S)) : KC_LDS_NONE;
    }
    const u32 splitA = (mode == KC_MODE_PAIR) ? lds_lower(b.A, 0, nA, S) : 0;
    const u32 splitB = (mode == KC_MODE_PAIR) ? lds_lower(b.B, 0, nB, S) : 0;
    for (u32 i = 0; i < nA; ++i) b.scr[i] = (i + 1 < nA) ? i + 1 : KC_LDS_NONE;
    for (u32 j = 0; j < nB; ++j) {
        u32 nx = (j + 1 < nB) ? j + 1 : KC_LDS_NONE;
        if (splitB > 0 && splitB < nB) {
            if (j == nB - 1) nx = 0;
            if (j == splitB - 1) nx = KC_LDS_NONE;
        }
        b.scr[nA + j] = nx;
    }
    u32 nev;
    if (g.lower_bound) {
        nev = mode == KC_MODE_PAIR ? kc_lds_replay<KC_MODE_PAIR, true>(g, b, nA, nB, splitA, splitB)
            : mode == KC_MODE_PAL ? kc_lds_replay<KC_MODE_PAL, true>(g, b, nA, nB, 0, 0) : kc_lds_replay<KC_MODE_UNI, true>(g, b, nA, nB, 0, 0);
    } else {
        nev = mode == KC_MODE_PAIR ? kc_lds_replay<KC_MODE_PAIR, false>(g, b, nA, nB, splitA, splitB)
            : mode == KC_MODE_PAL ? kc_lds_replay<KC_MODE_PAL, false>(g, b, nA, nB, 0, 0) : kc_lds_replay<KC_MODE_UNI, false>(g, b, nA, nB, 0, 0);
    }
    if (nev == 0) return;
    const u64 want = (u64)nev * (bi ? 2 : 1);
    u64 *logp;
#if defined(__HIP_DEVICE_COMPILE__)
    if (g.ne_sh) {   // sharded log (see ov_add_event)
        const u32 sh = (tag * 0x9E3779B1u) >> 26;
        const u64 pos = atomicAdd((unsigned long long *)&g.ne_shcur[sh * 16], (unsigned long long)want);
        const bool fits = pos + want <= g.ne_shcap;
        if (!fits) g.ne_shcur[64 * 16] = 1;
        logp = fits ? g.ne_sh + ((u64)sh * g.ne_shcap + pos) * 5 : nullptr;
    } else logp = g.ne + atomicAdd((unsigned long long *)g.ne_count, (unsigned long long)want) * 5;
#else
    (void)tag;
    logp = g.ne + *g.ne_count * 5; *g.ne_count += want;
#endif
    for (u32 e = 0; e < nev; ++e) {
        const u32 ev = b.evU[e], eidx = ev & 0x7fffffffu, ci = b.evC[e];
        const bool from2 = (ev >> 31) != 0;
        const u64 u = from2 ? lds_mirror(b.B[eidx], S) : b.A[eidx];
        const u64 c = from2 ? lds_mirror(b.A[ci], S) : b.B[ci];
        const u64 part = b.evPart[e];
        g.edge_from[u] = c; g.overlaps[u] = (u8)g.d; g.pred[c] = u;
        u64 *le = logp ? logp + (u64)e * (bi ? 2 : 1) * 5 : nullptr;
        if (le) { le[0] = u; le[1] = c; le[2] = u; le[3] = c; le[4] = part; }
        if (bi) {
            const u64 x2 = ov_mirror(c, g.S), y2 = ov_mirror(u, g.S);
            g.edge_from[x2] = y2; g.overlaps[x2] = (u8)g.d; g.pred[y2] = x2;
            if (le) { le[5] = x2; le[6] = y2; le[7] = u; le[8] = c; le[9] = part; }
        }
    }
}
