// TEST INFRASTRUCTURE ONLY: a sequential CPU replay of the phase-2 level loop of kmercamel_amd/csrc/kc_overlap.hip
// built on the SAME per-bucket logic header (kc_overlap_logic.h, compiled here for the host).  It lets the
// bucket-replay logic, the tagged path-end rule and the verify/repair loop be checked against the oracle without
// a GPU.  Buckets of a level are processed in a pseudo-random order to mimic concurrent buckets.
// Never linked into the product.
#include <vector>
#include <map>
#include <array>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include "../../kmercamel_amd/csrc/kc_overlap_logic.h"
#include "../../kmercamel_amd/csrc/kc_overlap_lds.h"   // the LDS replay (kc_lds_replay through kc_match_bucket_lane), compiled for the host

namespace {
template <int W> struct KLess { bool operator()(const Key<W> &a, const Key<W> &b) const { return key_lt(a, b); } };

template <int W>
int sim(const char **strs, u64 n, int k, int complements, int lower_bound, u64 *edge_from, u8 *overlaps, u64 seed, u64 *repairs_out, int lds_replay) {
    const u64 S = n, V = S * (complements ? 2 : 1);
    std::vector<Key<W>> pre(V), suf(V);
    auto enc = [&](const char *s, size_t off) { Key<W> x = key_zero<W>(); for (int i = 0; i < k; ++i) { int c = s[off + i]; u64 b = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; x = key_append(x, b, k); } return x; };
    for (u64 i = 0; i < S; ++i) {
        size_t L = strlen(strs[i]);
        Key<W> f = enc(strs[i], 0), l = enc(strs[i], L - k);
        pre[i] = f; suf[i] = l;
        if (complements) { pre[i + S] = key_rc(l, k); suf[i + S] = key_rc(f, k); }
    }
    std::vector<u64> pred(V, KC_NONE64), first(V), last(V), fs(V), ls(V), ne(V * 10 + 10), eidx(V, KC_NONE64), bl;
    std::vector<u8> cov(V, 0);
    for (u64 v = 0; v < V; ++v) { edge_from[v] = KC_NONE64; overlaps[v] = 0xff; fs[v] = ls[v] = v; }
    OvArgs g;
    g.S = S; g.V = V; g.batch = V / 16 + 1; g.complements = complements; g.lower_bound = lower_bound;
    g.edge_from = edge_from; g.overlaps = overlaps; g.pred = pred.data(); g.first = first.data(); g.last = last.data();
    g.first_snap = fs.data(); g.last_snap = ls.data(); g.ne = ne.data(); u64 ne_count = 0; g.ne_count = &ne_count; g.bl = nullptr; g.nbl = 0;
    u64 repairs = 0, rng = seed * 6364136223846793005ULL + 1442695040888963407ULL;
    for (int d = k - 1; d >= 0; --d) {
        g.d = d;
        std::map<Key<W>, std::pair<std::vector<u64>, std::vector<u64>>, KLess<W>> buckets;
        for (u64 v = 0; v < V; ++v) {
            if (pred[v] == KC_NONE64) buckets[key_prefix(pre[v], k, d)].second.push_back(v);
            if (edge_from[v] == KC_NONE64) buckets[key_suffix(suf[v], d)].first.push_back(v);
        }
        struct B { Key<W> key; const std::vector<u64> *A, *Bp; int mode; };
        std::vector<B> todo;
        for (auto &kv : buckets) {
            if (kv.second.first.empty() || kv.second.second.empty()) continue;
            int mode = KC_MODE_UNI;
            if (complements) {
                if (d == 0) mode = KC_MODE_PAL;
                else { Key<W> r = key_rc(kv.first, d); if (key_lt(r, kv.first)) continue; mode = key_eq(r, kv.first) ? KC_MODE_PAL : KC_MODE_PAIR; }
            }
            todo.push_back({kv.first, &kv.second.first, &kv.second.second, mode});
        }
        if (todo.empty()) continue;
        for (size_t i = todo.size(); i > 1; --i) { rng = rng * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(todo[i - 1], todo[(rng >> 33) % i]); }
        bl.clear(); g.bl = nullptr; g.nbl = 0;
        while (true) {
            ne_count = 0;
            if (!lower_bound) { std::fill(first.begin(), first.end(), ~0ull); std::fill(last.begin(), last.end(), ~0ull); }
            for (size_t t = 0; t < todo.size(); ++t) {
                if (lds_replay) {
                    // the replay the device runs out of LDS (one lane's form of it: the wavefront's form differs in who loads the
                    // bucket, not in the replay); every other bucket arrives in scrambled order, as in the grouped levels
                    std::vector<u64> A(*todo[t].A), B(*todo[t].Bp);
                    const bool unsorted = (t & 1) != 0;
                    if (unsorted) {
                        for (size_t i = A.size(); i > 1; --i) { rng = rng * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(A[i - 1], A[(rng >> 33) % i]); }
                        for (size_t i = B.size(); i > 1; --i) { rng = rng * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(B[i - 1], B[(rng >> 33) % i]); }
                    }
                    std::vector<unsigned char> mem(kc_lds_bytes(A.size(), B.size()) + 64);
                    kc_match_bucket_lane(g, A.data(), (u32)A.size(), B.data(), (u32)B.size(), todo[t].mode, mem.data(), unsorted, (u32)t);
                    continue;
                }
                std::vector<u64> sA(todo[t].A->size() + 1), sB(todo[t].Bp->size() + 1);
                kc_match_bucket(g, todo[t].A->data(), todo[t].A->size(), todo[t].Bp->data(), todo[t].Bp->size(), todo[t].mode, (u32)t, sA.data(), sB.data());
            }
            u64 nNew = ne_count;
            if (nNew == 0 || lower_bound) break;
            u64 covered = 0;
            for (u64 e = 0; e < nNew; ++e) {   // ov_fix_kernel<false>
                u64 y = ne[e * 5 + 1], l0 = ls[y];
                if (edge_from[l0] != KC_NONE64) continue;
                u64 f = y, steps = 0;
                while (steps <= nNew) { u64 xp = pred[f]; if (xp == KC_NONE64) break; cov[f] = 1; ++steps; f = fs[xp]; }
                covered += steps;
            }
            if (covered == nNew) {
                for (u64 e = 0; e < nNew; ++e) {   // ov_fix_kernel<true>
                    u64 y = ne[e * 5 + 1], l0 = ls[y];
                    if (edge_from[l0] != KC_NONE64) continue;
                    u64 f = y, steps = 0;
                    while (steps <= nNew) { u64 xp = pred[f]; if (xp == KC_NONE64) break; ++steps; f = fs[xp]; }
                    fs[l0] = f; ls[f] = l0;
                }
                for (u64 e = 0; e < nNew; ++e) cov[ne[e * 5 + 1]] = 0;
                break;
            }
            ++repairs;
            for (u64 e = 0; e < nNew; ++e) eidx[ne[e * 5 + 1]] = e;
            u64 bp = ~0ull, bu = ~0ull, bc = 0; bool found = false;
            for (u64 e = 0; e < nNew; ++e) {   // ov_find_cycle_kernel
                u64 y0 = ne[e * 5 + 1];
                if (cov[y0]) continue;
                u64 cur = y0, mp = 0, mu = 0, mc = 0; bool any = false;
                do {
                    cov[cur] = 1;
                    const u64 *ed = &ne[eidx[cur] * 5];
                    if (!any || ed[4] > mp || (ed[4] == mp && ed[2] > mu)) { mp = ed[4]; mu = ed[2]; mc = ed[3]; any = true; }
                    cur = fs[ed[0]];
                } while (cur != y0);
                if (!found || mp < bp || (mp == bp && mu < bu)) { bp = mp; bu = mu; bc = mc; found = true; }
            }
            if (!found) return -1;
            bl.push_back(bu); bl.push_back(bc);
            g.bl = bl.data(); g.nbl = bl.size() / 2;
            for (u64 e = 0; e < nNew; ++e) { u64 x = ne[e * 5], y = ne[e * 5 + 1]; edge_from[x] = KC_NONE64; overlaps[x] = 0xff; pred[y] = KC_NONE64; eidx[y] = KC_NONE64; cov[y] = 0; }
            if (repairs > 100000) return -2;
        }
    }
    if (repairs_out) *repairs_out = repairs;
    return 0;
}
}  // namespace

extern "C" int sim_overlap_path2(const char **strs, uint64_t n, int k, int complements, int lower_bound, uint64_t *edge_from, uint8_t *overlaps,
                                 uint64_t seed, uint64_t *repairs, int lds_replay) {
    switch (kc_words_for_k(k)) {
        case 1: return sim<1>(strs, n, k, complements, lower_bound, edge_from, overlaps, seed, repairs, lds_replay);
        case 2: return sim<2>(strs, n, k, complements, lower_bound, edge_from, overlaps, seed, repairs, lds_replay);
        default: return sim<4>(strs, n, k, complements, lower_bound, edge_from, overlaps, seed, repairs, lds_replay);
    }
}
extern "C" int sim_overlap_path(const char **strs, uint64_t n, int k, int complements, int lower_bound, uint64_t *edge_from, uint8_t *overlaps,
                                uint64_t seed, uint64_t *repairs) {
    return sim_overlap_path2(strs, n, k, complements, lower_bound, edge_from, overlaps, seed, repairs, 0);
}
