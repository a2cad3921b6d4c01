// Sequential-equivalent greedy matching inside one key bucket of one overlap length d.
// Restates the inner loops of OverlapHamiltonianPath (global.h:63-128) in a layout-free form:
//
//   reference event order      : for d = k-1..0, for part = 0..15, for suffix vertex i ascending (global.h:63,66,86)
//   candidates of an event      : prefix vertices j with prefix_d(j) == suffix_d(i) inside part's index range
//                                 [part*batch, (part+1)*batch), batch = V/16+1, in DESCENDING j (the per-key chains are
//                                 built by head insertion, global.h:73-85), skipping j that already has an in-edge,
//                                 j == mirror(i) and j == first vertex of i's path (global.h:91-105)
//   accepted edge               : i -> j with overlap d, plus the mirrored edge mirror(j) -> mirror(i) (global.h:108-122)
//
// Vertices with the same d-mer key only interact with each other and -- through the mirrored edge -- with the bucket
// of the reverse-complement key, whose members are exactly the mirrors of this bucket's members.  One call of
// kc_match_bucket therefore replays, in the reference's order, all events of the bucket pair {key, RC(key)}.
// The cycle test needs the first vertex of i's path.  Buckets of one level run concurrently and another bucket may be
// ahead in the reference's event order, so a bucket must not read path ends written by other buckets in this level:
// live path-end entries carry the id of the bucket that wrote them; a bucket trusts only its own entries and
// otherwise falls back to the level-start snapshot.  The information it uses is therefore always a sub-forest of
// the true path forest at that event, so the test can only MISS a cycle (never reject a legal edge); a missed cycle
// shows up as a cycle after the level, which the caller detects and repairs (DESIGN.md "Exactness of phase 2").
//
// Compiles for the device (hipcc) and for the host (g++, used by tests/sim to check this logic without a GPU).
#pragma once
#include "kc_key.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define KC_LOG_APPEND(ptr) atomicAdd((unsigned long long *)(ptr), 1ULL)
#else
#define KC_LOG_APPEND(ptr) ((*(ptr))++)
#endif

struct OvArgs {
    u64 S, V, batch;
    int d, complements, lower_bound;
    u64 *edge_from; u8 *overlaps; u64 *pred;      // live state, V entries each
    u64 *first; u64 *last;                        // live path ends of this level: (writer tag << 32) | vertex
    const u64 *first_snap; const u64 *last_snap;  // exact path ends at the start of the level
    u64 *ne;            // new-edge log: 5 words per edge {x, y, event_u, event_c, part}
    u64 *ne_count;
    u64 *ne_sh;         // sharded new-edge log (device replay only; null: log straight into ne): 64 shards of ne_shcap entries, cursors 16 words apart
    u64 *ne_shcur; u64 ne_shcap;
    const u64 *abort2;  // two device words (or null): non-zero = the sorted hit lists are void (a bucketed sort overflowed): do nothing
    const u64 *bl;      // blacklist of (u, c) pairs proven to close a cycle
    u64 nbl;
    u32 tag_base = 0;   // first writer tag of the current launch (tags are never reused between launches)
    u32 lds_min = 12;   // buckets with at least this many members are replayed out of LDS by a wavefront of their own
    u32 lane_off = 0;   // 1: no lane-per-bucket LDS replay of the small buckets (kc_match_bucket_lane): they take kc_match_bucket
#ifdef KC_EXPERIMENTS
    u32 exp = 0;        // timing experiments (tools' build only): 1 = ov_match_kernel stops after finding its bucket
#endif
};

#define KC_MODE_UNI 0
#define KC_MODE_PAL 1
#define KC_MODE_PAIR 2

struct OvList {      // ascending list of vertex ids; `mirrored` views base[] through v -> mirror(v), still ascending
    const u64 *base;
    u64 n, split, S;
    int mirrored;
    KC_HD u64 get(u64 t) const {
        if (!mirrored) return base[t];
        u64 hi = n - split;  // elements >= S come first after mirroring (v - S), then the ones < S (v + S)
        return t < hi ? base[split + t] - S : base[t - hi] + S;
    }
};
KC_HD u64 ov_lower_bound(const OvList &l, u64 v) {  // first t with l.get(t) >= v
    u64 lo = 0, hi = l.n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (l.get(mid) < v) lo = mid + 1; else hi = mid; }
    return lo;
}
KC_HD u64 ov_split(const u64 *a, u64 n, u64 S) {  // first index with a[idx] >= S
    u64 lo = 0, hi = n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] < S) lo = mid + 1; else hi = mid; }
    return lo;
}
KC_HD u64 ov_mirror(u64 v, u64 S) { return v < S ? v + S : v - S; }

#define KC_NO_TAG 0xffffffffu
KC_HD u64 ov_end_of(const u64 *live, const u64 *snap, u64 v, u32 tag) {
    u64 w = live[v];
    return ((u32)(w >> 32) == tag) ? (w & 0xffffffffull) : snap[v];
}

// Accept the event u -> c (and, bidirectional, the mirrored edge mirror(c) -> mirror(u)), global.h:108-122.
// All path-end loads of both edges are independent of each other and of this event's stores (a path never contains
// its own mirror), so they are issued together: one memory round trip instead of six.
KC_HD void ov_add_event(const OvArgs &g, u64 u, u64 c, u64 firstU, u64 part, u32 tag) {
    const u64 S = g.S;
    const bool two = g.complements != 0;
    const u64 x2 = two ? ov_mirror(c, S) : 0, y2 = two ? ov_mirror(u, S) : 0;
    u64 lastC = 0, firstX2 = 0, lastY2 = 0;
    if (!g.lower_bound) {
        u64 l1 = g.last[c], s1 = g.last_snap[c];
        u64 l2 = two ? g.first[x2] : 0, s2 = two ? g.first_snap[x2] : 0;
        u64 l3 = two ? g.last[y2] : 0, s3 = two ? g.last_snap[y2] : 0;
        lastC = ((u32)(l1 >> 32) == tag) ? (l1 & 0xffffffffull) : s1;
        firstX2 = ((u32)(l2 >> 32) == tag) ? (l2 & 0xffffffffull) : s2;
        lastY2 = ((u32)(l3 >> 32) == tag) ? (l3 & 0xffffffffull) : s3;
    }
    u64 pos = 0;
    u64 *e = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if (g.ne_sh) {
        // one add per event on ONE word serialises at ~12 ns (5e5 events of a level: several ms): the log is written through 64
        // cursors (the bucket's tag picks one) and gathered into the dense log afterwards (ov_gather_log_kernel)
        const u32 sh = (tag * 0x9E3779B1u) >> 26;
        pos = atomicAdd((unsigned long long *)&g.ne_shcur[sh * 16], two ? 2ULL : 1ULL);
        e = pos + 2 <= g.ne_shcap ? g.ne_sh + ((u64)sh * g.ne_shcap + pos) * 5 : nullptr;
        if (!e) g.ne_shcur[64 * 16] = 1;   // a shard ran full (cannot happen with the caller's sizing): the caller sees the flag
    } else {
        pos = atomicAdd((unsigned long long *)g.ne_count, two ? 2ULL : 1ULL);
        e = g.ne + pos * 5;
    }
#else
    pos = *g.ne_count; *g.ne_count += two ? 2 : 1;
    e = g.ne + pos * 5;
#endif
    g.edge_from[u] = c; g.overlaps[u] = (u8)g.d; g.pred[c] = u;
    if (!g.lower_bound) {
        g.first[lastC] = ((u64)tag << 32) | firstU;
        g.last[firstU] = ((u64)tag << 32) | lastC;
    }
    if (!e) return;
    e[0] = u; e[1] = c; e[2] = u; e[3] = c; e[4] = part;
    if (two) {
        g.edge_from[x2] = y2; g.overlaps[x2] = (u8)g.d; g.pred[y2] = x2;
        if (!g.lower_bound) {
            g.first[lastY2] = ((u64)tag << 32) | firstX2;
            g.last[firstX2] = ((u64)tag << 32) | lastY2;
        }
        e[5] = x2; e[6] = y2; e[7] = u; e[8] = c; e[9] = part;
    }
}

// A: suffix-side members (vertex ids ascending) of the bucket, B: prefix-side members (ascending).
// sA / sB: scratch of nA / nB words owned by this bucket: after the first part it holds the (shrinking) ordered list of
// still unmatched event vertices, so later parts do not re-scan the matched ones.
KC_HD void kc_match_bucket(const OvArgs &g, const u64 *A, u64 nA, const u64 *B, u64 nB, int mode, u32 tag, u64 *sA, u64 *sB) {
    const u64 S = g.S;
    OvList E1{A, nA, 0, S, 0}, C1{B, nB, 0, S, 0};
    OvList E2{B, 0, 0, S, 1}, C2{A, 0, 0, S, 1};
    if (mode == KC_MODE_PAIR) {
        E2.n = nB; E2.split = ov_split(B, nB, S);
        C2.n = nA; C2.split = ov_split(A, nA, S);
    }
    // every event vertex is an active (unmatched) suffix at level start; an accepted event consumes u and, in
    // bidirectional mode, mirror(c), which is an event vertex of this bucket pair as well
    u64 left = E1.n + E2.n, cur_n = 0;
    bool use_scratch = false;
    // parts (candidate index ranges of V/16+1 vertices, global.h:63-72) without a candidate of this bucket do nothing in the
    // reference either: go from one non-empty part to the next by following the (ascending) candidate lists
    u64 nx1 = 0, nx2 = 0;   // first candidate of C1 / C2 not yet assigned to a part
    while (left > 0) {
        const u64 v1 = nx1 < C1.n ? C1.get(nx1) : ~(u64)0, v2 = nx2 < C2.n ? C2.get(nx2) : ~(u64)0;
        const u64 vmin = v1 < v2 ? v1 : v2;
        if (vmin == ~(u64)0) break;
        const u64 part = vmin / g.batch, hi_v = (part + 1) * g.batch;
        const u64 cl1 = nx1, cl2 = nx2;
        while (nx1 < C1.n && C1.get(nx1) < hi_v) ++nx1;
        while (nx2 < C2.n && C2.get(nx2) < hi_v) ++nx2;
        u64 top1 = nx1, top2 = nx2;
        u64 i1 = 0, i2 = 0, si = 0, out_n = 0;
        bool exhausted = false;   // every candidate of this part is used: no later event of the part can match
        while (use_scratch ? (si < cur_n) : (i1 < E1.n || i2 < E2.n)) {
            if (exhausted && use_scratch) {   // the rest of the list is carried over as it is (moved down behind the kept events)
                for (; si < cur_n; ++si, ++out_n) {
                    if (out_n == si) continue;
                    const u64 v = si < nA ? sA[si] : sB[si - nA];
                    if (out_n < nA) sA[out_n] = v; else sB[out_n - nA] = v;
                }
                break;
            }
            u64 u; int from;
            if (use_scratch) {
                u64 v = si < nA ? sA[si] : sB[si - nA];
                ++si;
                u = v >> 1; from = (v & 1) ? 2 : 1;
            } else if (i2 >= E2.n) { u = E1.get(i1++); from = 1; }
            else if (i1 >= E1.n) { u = E2.get(i2++); from = 2; }
            else {
                u64 a = E1.get(i1), b = E2.get(i2);
                if (a <= b) { u = a; ++i1; from = 1; } else { u = b; ++i2; from = 2; }
            }
            // independent loads of the event, issued together
            const u64 ef = g.edge_from[u];
            const u64 fl = g.lower_bound ? 0 : g.first[u], fsn = g.lower_bound ? 0 : g.first_snap[u];
            if (ef != KC_NONE64) continue;                                   // suffixForbidden (global.h:87)
            if (exhausted) {   // first part: just list the event
                u64 v = (u << 1) | (u64)(from == 2);
                if (out_n < nA) sA[out_n] = v; else sB[out_n - nA] = v;
                ++out_n;
                continue;
            }
            const u64 firstU = ((u32)(fl >> 32) == tag) ? (fl & 0xffffffffull) : fsn;
            bool matched = false;
            const OvList &C = (from == 1) ? C1 : C2;
            u64 &top = (from == 1) ? top1 : top2;
            const u64 cl = (from == 1) ? cl1 : cl2;
            while (top > cl && g.pred[C.get(top - 1)] != KC_NONE64) --top;    // used candidates never come back
            if (top == cl) {   // this event's candidates are gone; if the other list's are too, the part is over
                while (top1 > cl1 && g.pred[C1.get(top1 - 1)] != KC_NONE64) --top1;
                while (top2 > cl2 && g.pred[C2.get(top2 - 1)] != KC_NONE64) --top2;
                exhausted = (top1 == cl1 && top2 == cl2);
            }
            for (u64 t = top; t > cl; --t) {
                u64 c = C.get(t - 1);
                if (g.pred[c] != KC_NONE64) continue;                         // prefixForbidden (global.h:97)
                if (!g.lower_bound) {
                    if (g.complements && c == ov_mirror(u, S)) continue;      // complementary (global.h:93)
                    if (firstU == c) continue;                                // would close a cycle (global.h:95)
                    bool banned = false;
                    for (u64 q = 0; q < g.nbl; ++q) banned = banned || (g.bl[2 * q] == u && g.bl[2 * q + 1] == c);
                    if (banned) continue;
                }
                ov_add_event(g, u, c, firstU, part, tag);
                left -= 1;
                if (g.complements && (!g.lower_bound || ov_mirror(c, S) != u)) left -= (left > 0);
                matched = true;
                break;
            }
            if (!matched) {   // keep for the next part (written behind the read position: in-place safe)
                u64 v = (u << 1) | (u64)(from == 2);
                if (out_n < nA) sA[out_n] = v; else sB[out_n - nA] = v;
                ++out_n;
            }
        }
        use_scratch = true;
        cur_n = out_n;
    }
}
