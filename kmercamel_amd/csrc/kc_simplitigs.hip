// Phase 1 of the BIGREEDY path on gfx950: greedy simplitigs by de Bruijn neighbour lookup.
// Replaces simplitig_right_rev_extension / next_simplitig / get_simplitigs (simplitigs.h:105-205) and
// simplitigs_from_fasta (simplitigs.h:78-103).
//
// The reference walks the hash table sequentially, deleting k-mers as it goes, so its result depends on the khash
// bucket layout.  Here the same object -- a vertex-disjoint path cover of the node-centric de Bruijn
// graph in which no two path ends could still be joined -- is computed by a scheduling-free rule (DESIGN.md
// "Canonical simplitigs", restated on the CPU in oracle/kc_oracle.cpp simplitigs_canonical_order):
//   * every k-mer has two ports, L (prepend side) and R (append side); an edge joins two ports;
//   * edges are matched greedily by a symmetric 64-bit priority (locally dominant edges, iterated to a fixpoint:
//     every open port picks its best still-available edge, an edge is accepted when both of its ports picked it);
//   * chains start at the end with the smaller key and are ordered by that key; cycles are cut at their minimum key.
// Kernels: kc_choose_dense_kernel (round 1: 8 batched lookups per k-mer, block tags staged in LDS), kc_reject / kc_choose /
// kc_accept (later rounds), kc_walk_measure (walk + 2-bit recording), kc_stream_emit, the segment-parallel kernels for long
// chains and the cycle cutters; host flow in build_simplitigs_w.
#include "kc_internal.h"

#include "kc_simp_dev.h"

// node ids leave gaps behind every block (kc_set_dev.h): the first ncount[b] of block b's nstride ids are in use
// the slot of a listed port's k-mer (see KC_WK_TAG)
template <int W> __device__ __forceinline__ u64 kc_wk_slot(const SetView<W> &set, u64 e) {
    const u64 nid = KC_WK_PID(e) >> 1;
    if (!(e & KC_WK_HAS_OFF)) return set_slot_of_nid(set, nid);
    return ((nid / set.nstride) << set.bshift) | ((e >> 40) & (((u64)1 << set.bshift) - 1));
}
__device__ __forceinline__ bool kc_nid_in_use(u64 nid, const u32 *__restrict__ ncount, u32 nstride) {
    const u64 blk = nid / nstride;
    return (u32)(nid - blk * nstride) < ncount[blk];
}
// Branch ports re-enumerate their candidates and reject every existing one they did not choose: the rejected port's
// round-1 choice (which can only have been this port) is not reciprocated, so it joins the work list of round 2.
template <int W>
__global__ void __launch_bounds__(256) kc_reject_kernel(SetView<W> set, u64 *link, int k, int complements, const u64 *__restrict__ branch,
                                                        u64 nbranch, u64 *work_out, u64 work_cap, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 rej[4];   // the ports this one rejects (entries for the work list of round 2)
    u32 nrej = 0;
    if (t < nbranch) {
        const u64 pid = KC_WK_PID(branch[t]);   // (ports are numbered by dense node id: kc_set_dev.h)
        const u64 slot = kc_wk_slot<W>(set, branch[t]);
        const int port = (int)(pid & 1);
        const Key<W> x = set_key_at(set, slot);
        const Key<W> xr = key_rc(x, k);
        const u64 chosen = LK_PID(link[pid]);
        // the candidates of a port share their home (the middle): ONE window search finds all of them, as in round 1
        u64 cs[4]; u32 qb;
        bool ok = false;
        if (set.m) {
            const Key<W> lm = key_lowmask<W>(2 * (k - 2));
            const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
            const KcPortDesc d = kc_port_desc<W, 0>(set, x, xr, k, complements, port, kc_mid_min<W, 0>(mid, midr, k, set.m));
            ok = kc_port_search<W, false, false>(set, k, complements, x, xr, slot, port, d, nullptr, nullptr, cs, qb);
        }
        if (!ok) kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
#pragma unroll
        for (u32 c = 0; c < 4; ++c) {
            const u64 tslot = cs[c];
            if (tslot == KC_NONE64) continue;
            const u64 cpid = (set_nid_of_slot(set, tslot) << 1) | (u64)((qb >> c) & 1u);
            if (cpid == chosen) continue;
            const u64 wc = link[cpid];   // round-1 words are stable here apart from their LK_REJ bit
            if (LK_STATE(wc) == LK_CHOICE && LK_PID(wc) == pid) {   // it chose this port, this port chose another: void, round 2
                atomicOr((ull *)&link[cpid], (ull)LK_REJ);           // (only its target can reject a port, so this happens once)
                const u64 e = cpid | KC_WK_TAG(tslot & (((u64)1 << set.bshift) - 1));
                if (nrej == 0) rej[0] = e; else if (nrej == 1) rej[1] = e; else if (nrej == 2) rej[2] = e; else rej[3] = e;
                ++nrej;
            }
        }
    }
    // one reservation per workgroup (reads with errors list 2 10^7 ports here: one add per port -- per wave and loop trip after the
    // compiler's aggregation -- on one word took most of this kernel's 30 ms)
    u64 pos = kc_block_reserve(nrej, cursor);
    if (nrej > 0) { if (pos < work_cap) work_out[pos] = rej[0]; ++pos; }
    if (nrej > 1) { if (pos < work_cap) work_out[pos] = rej[1]; ++pos; }
    if (nrej > 2) { if (pos < work_cap) work_out[pos] = rej[2]; ++pos; }
    if (nrej > 3) { if (pos < work_cap) work_out[pos] = rej[3]; }
}

// One thread per port.  FIRST: port id = thread id over all 2*cap ports in XCD-aware order (round 1, nothing is matched
// yet, so the neighbours' link words are not read); otherwise the port ids come from the work list of unresolved ports
// (later rounds: few ports, one window search each -- the four candidates of a port share their home).
template <int W, bool FIRST, int KC>
__global__ void __launch_bounds__(256) kc_choose_kernel(SetView<W> set, u64 *__restrict__ link, int k_rt, int complements,
                                                        const u64 *__restrict__ work, u64 nwork, int xcd_aware, u64 *heads,
                                                        u64 heads_cap, u64 *heads_cursor) {
    const int k = KC ? KC : k_rt;
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    (void)xcd_aware;
    if (t >= nwork) return;
    static_assert(!FIRST, "round 1 is kc_probe_block_kernel / kc_choose_dense_kernel");
    u64 pid = KC_WK_PID(work[t]);   // (dense node id << 1 | port)
    u64 slot = kc_wk_slot<W>(set, work[t]);
    int port = (int)(pid & 1);
    Key<W> x = set_key_at(set, slot);
    Key<W> xr = key_rc(x, k);
    if (complements && port == 1 && key_eq(x, xr)) {   // palindromes link on port L only
        link[pid] = LK_MAKE(LK_DEAD, 0, 0, slot) | (link[pid] & LK_SPLIT);
        kc_push_head(pid, complements, heads, heads_cap, heads_cursor);
        return;
    }
    const u64 hxk = key_hash(x);
    int nfound = 0;
    Key<W> by = x; int bq = 0; u64 bpr = 0, bslot = 0; u32 bbase = 0;
    // (one window search for the four candidates of the port: they share their home)
    u64 cs[4]; u32 qb;
    bool ok = false;
    if (set.m) {
        const Key<W> lm = key_lowmask<W>(2 * (k - 2));
        const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
        const KcPortDesc d = kc_port_desc<W, 0>(set, x, xr, k, complements, port, kc_mid_min<W, 0>(mid, midr, k, set.m));
        ok = kc_port_search<W, false, false>(set, k, complements, x, xr, slot, port, d, nullptr, nullptr, cs, qb);
    }
    if (!ok) kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
    for (u32 c = 0; c < 4; ++c) {
        u64 tslot = KC_NONE64;
#pragma unroll
        for (u32 cc = 0; cc < 4; ++cc) tslot = (c == cc) ? cs[cc] : tslot;
        if (tslot == KC_NONE64) continue;
        int q;
        const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
        tslot = set_nid_of_slot(set, tslot);   // from here on: the candidate's dense id
        {
            u64 wt = link[(tslot << 1) | (u64)q];
            if (LK_IS_MATCHED(wt)) continue;
        }
        // priorities are only needed to choose between two existing candidates
        if (nfound == 0) { by = y; bq = q; bslot = tslot; bbase = port ? c : 3 - c; nfound = 1; continue; }
        if (nfound == 1) { bpr = kc_mix64(kc_port_term(hxk, port) + kc_port_term(key_hash(by), bq)); nfound = 2; }
        u64 pr = kc_mix64(kc_port_term(hxk, port) + kc_port_term(key_hash(y), q));
        if (edge_preferred(x, port, y, q, pr, by, bq, bpr)) { by = y; bq = q; bpr = pr; bslot = tslot; bbase = port ? c : 3 - c; }
    }
    // FIRST is no longer launched (round 1 = kc_choose_dense_kernel); later rounds mark their choice as tentative
    link[pid] = (nfound ? (LK_MAKE(LK_CHOICE, bq, bbase, bslot) | LK_REJ) : LK_MAKE(LK_DEAD, 0, 0, slot)) | (link[pid] & LK_SPLIT);
    if (!nfound) kc_push_head(pid, complements, heads, heads_cap, heads_cursor);
}

// Fallback for heavily branching graphs (the branch list overflowed): check every round-1 choice against its target
// directly. FILL = false counts the unreciprocated ports, FILL = true marks them LK_REJ and lists them.
template <bool FILL>
__global__ void __launch_bounds__(256) kc_mutual_scan_kernel(u64 *link, u64 n_nodes, const u32 *__restrict__ ncount, u32 nstride, u64 *work_out, u64 *cursor) {
    // a thread takes 32 node ids (the ids in use: the first ncount[b] of every block's nstride)
    const u64 n0 = kc_gtid() * 32;
    u64 m = 0;   // bit 2i+p: port p of node n0+i is an unreciprocated choice
    for (int i = 0; i < 32 && n0 + (u64)i < n_nodes; ++i)
        for (int port = 0; port < 2; ++port) {
            if (!kc_nid_in_use(n0 + (u64)i, ncount, nstride)) continue;
            const u64 pid = ((n0 + (u64)i) << 1) | (u64)port;
            u64 w = link[pid];
            if (LK_STATE(w) != LK_CHOICE) continue;
            u64 wt = link[LK_PID(w)];   // only LK_REJ bits change concurrently
            if (LK_STATE(wt) == LK_CHOICE && LK_PID(wt) == pid) continue;
            m |= (u64)1 << (2 * i + port);
        }
    u64 pos = kc_block_reserve((u32)__popcll(m), cursor);   // (branching inputs list 10^7 ports here: one add per workgroup, not per port)
    if (FILL)
        for (int j = 0; m != 0 && j < 64; ++j)
            if ((m >> j) & 1) { const u64 pid = (n0 << 1) + (u64)j; atomicOr((ull *)&link[pid], (ull)LK_REJ); work_out[pos++] = pid; }
}

// Later rounds: a tentative choice (CHOICE|REJ) is confirmed when its target's tentative choice points back.
__global__ void __launch_bounds__(256) kc_accept_kernel(u64 *link, const u64 *__restrict__ work, u64 nwork, u64 *work_out,
                                                        u64 work_cap, u64 *cursor) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 pid = 0, e = 0;
    bool again = false;   // still unresolved: goes on the next round's list
    if (t < nwork) {
        e = work[t];
        pid = KC_WK_PID(e);
        u64 w = link[pid];
        if (LK_STATE(w) == LK_CHOICE && (w & LK_REJ)) {
            u64 wt = __hip_atomic_load(&link[LK_PID(w)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool back = LK_PID(wt) == pid && ((LK_STATE(wt) == LK_CHOICE && (wt & LK_REJ)) || LK_STATE(wt) == LK_MATCHED);
            if (back) __hip_atomic_store(&link[pid], (w & ~(3ull | LK_REJ)) | LK_MATCHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else again = true;
        }
    }
    const u64 pos = kc_block_reserve(again ? 1u : 0u, cursor);
    if (again && pos < work_cap) work_out[pos] = e;   // (with its slot offset, if it came with one)
}

// heads of directed chains: oriented node (id, o) whose entry port (L for o=0, R for o=1) is not matched.
template <bool FILL>
__global__ void __launch_bounds__(256) kc_heads_kernel(const u64 *__restrict__ link, u64 n_nodes, const u32 *__restrict__ ncount, u32 nstride, int complements, u64 *heads, u64 *cursor) {
    const u64 n0 = kc_gtid() * 32;   // 32 node ids per thread
    u64 m = 0;   // bit 2i+o: (n0+i, o) is a head
    const int no = complements ? 2 : 1;
    for (int i = 0; i < 32 && n0 + (u64)i < n_nodes; ++i)
        for (int o = 0; o < no; ++o) {
            if (!kc_nid_in_use(n0 + (u64)i, ncount, nstride)) continue;
            u64 w = link[((n0 + (u64)i) << 1) | (u64)o];
            if (!LK_IS_MATCHED(w)) m |= (u64)1 << (2 * i + o);
        }
    u64 pos = kc_block_reserve((u32)__popcll(m), cursor);
    if (FILL)
        for (int j = 0; m != 0 && j < 64; ++j)
            if ((m >> j) & 1) heads[pos++] = (n0 << 1) + (u64)j;
}

#define KC_WALK_CLASSES 4
__device__ __forceinline__ u32 kc_walk_class(u64 head) { return (u32)(kc_mix64(head) >> 20) & (KC_WALK_CLASSES - 1); }
// the heads dealt into their classes: out[c * n ...] = the heads of class c (any order), counts[c] of them.  One reservation per class
// and WORKGROUP of 1024 (a reservation per wave was 3.8 ms of same-address atomics for 5 * 10^6 heads)
__global__ void __launch_bounds__(1024) kc_heads_by_class_kernel(const u64 *__restrict__ heads, u64 n, u64 *out, u64 *counts) {
    __shared__ u32 wg_cnt[KC_WALK_CLASSES];
    __shared__ u64 wg_base[KC_WALK_CLASSES];
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = t < n;
    const u64 h = in ? heads[t] : 0;
    const u32 c = in ? kc_walk_class(h) : KC_WALK_CLASSES;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < KC_WALK_CLASSES) wg_cnt[threadIdx.x] = 0;
    __syncthreads();
    u32 mine = 0;   // this lane's place among the workgroup's heads of its class
#pragma unroll
    for (u32 q = 0; q < KC_WALK_CLASSES; ++q) {
        const unsigned long long m = __ballot(c == q);
        if (!m) continue;
        const int leader = __builtin_ctzll(m);
        u32 base = 0;
        if (lane == leader) base = atomicAdd(&wg_cnt[q], (u32)__popcll(m));
        base = (u32)__shfl((int)base, leader, 64);
        if (c == q) mine = base + (u32)__popcll(m & (((unsigned long long)1 << lane) - 1));
    }
    __syncthreads();
    if (threadIdx.x < KC_WALK_CLASSES && wg_cnt[threadIdx.x]) wg_base[threadIdx.x] = atomicAdd((ull *)&counts[threadIdx.x], (ull)wg_cnt[threadIdx.x]);
    __syncthreads();
    if (in) out[(u64)c * n + wg_base[c] + (u64)mine] = h;
}
#define LK_WALKED (1ull << 62)          /* on a dead port: the chain that ends here was walked from its other end (kc_walk_measure_kernel) */
#define KC_STREAM_TURNED (1ull << 63)   /* chain_stream: the recording runs from the chain's LAST k-mer: the emit writes its reverse complement */
// the class passes were voided (a chain beyond the limit, cycles): the marks they left on dead ports go away with them, so that bit 62 of
// a link word means nothing outside kc_walk_measure_kernel's own launches
static __global__ void __launch_bounds__(256) kc_clear_walked_kernel(u64 *link, const u64 *__restrict__ heads, u64 n) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { const u64 w = link[heads[t]]; if (w & LK_WALKED) link[heads[t]] = w & ~LK_WALKED; }
}
template <int W, bool MARK, bool REC, bool FRAG = false>
__global__ void __launch_bounds__(256) kc_walk_measure_kernel(SetView<W> set, const u64 *link, int complements,   // (no __restrict__: link_marks may alias link)
                                                              const u64 *__restrict__ heads, u64 nheads, u8 *visited,
                                                              u64 *chain_head, u64 *chain_len, u64 *chain_key, u64 *cursor, u64 limit, KcRec rec,
                                                              KcFrag frag = KcFrag{nullptr, nullptr, 0, 0}, u64 *link_marks = nullptr, int cls_pass = 0,
                                                              u64 t0 = 0, u64 nheads_all = 0) {
    // limit: give up (cursor[7] = 1) on a chain longer than this; the caller then switches to the segment-parallel path
    // Nodes are dense ids.  The two places that need a KEY -- the chain's two end k-mers -- find its slot in the word of the dead
    // port there: the head's entry port, and the exit port the walk ends on.
    // link_marks != nullptr (bidirectional sets, round 5): ONE WALK PER CHAIN where possible.  Both ends of a chain are heads, and a walk
    // reads the whole chain's link words -- the kernel moves 130 GB at HBM speed for 2.5 * 10^9 k-mers, half of it for the walks whose
    // recording is thrown away because the other end has the smaller key.  The heads are dealt into KC_WALK_CLASSES classes by a hash of
    // their id and the kernel runs once per class: a walk whose far end belongs to a LATER class marks that end's dead port (LK_WALKED in
    // link_marks = the link words) and the far end never starts; it keeps its recording whichever end has the smaller key, and says so
    // (bit 63 of chain_stream: the emit turns the recording round).  Two ends of the same class both walk, as before: 1 + 1 / classes
    // walks per chain.  `heads` is then the list of ONE class (kc_heads_by_class_kernel: full waves), t0 its offset among all
    // nheads_all heads -- walk numbers, i.e. first chunks, stay unique over the launches.
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nheads) return;
    u64 h = heads[t];
    if (link_marks && cls_pass && (link_marks[h] & LK_WALKED)) return;   // the chain was walked from its other end
    t += t0;
    if (!nheads_all) nheads_all = nheads;
    u64 cur = h >> 1;
    int o = (int)(h & 1), co = o;
    u64 len = 1;
    u64 acc = 0;
    int nacc = 0, wi = 0;
    u32 chunk = KC_NO_CHUNK, first_chunk = KC_NO_CHUNK;
    bool rec_ok = true;
    auto flush_word = [&](u64 word) {
        if (chunk == KC_NO_CHUNK) {
            // the first chunk of walk t is chunk t: no reservation.  (Lanes of a wave reach their first flush at different loop
            // trips once walks hop over fragments, so a shared cursor here meant ~one same-address atomic per walk, ~13 ns each.)
            chunk = first_chunk = (u32)t;
            wi = 0;
        } else if (wi == KC_CHUNK_WORDS) {
            const u64 nc = nheads_all + atomicAdd((ull *)rec.pool_cursor, 1ULL);
            if (nc >= rec.pool_cap) { rec_ok = false; return; }
            rec.chunk_next[chunk] = (u32)nc;
            chunk = (u32)nc;
            wi = 0;
        }
        rec.pool[(u64)chunk * KC_CHUNK_WORDS + (u64)wi++] = word;
    };
    // append n (1..32) bases given right-aligned in `bits` (first base in the highest pair) to the recording
    auto push_group = [&](u64 bits, int n) {
        const int room = 32 - nacc;
        if (n < room) { acc = (acc << (2 * n)) | bits; nacc += n; return; }
        const int rest = n - room;                                   // `room` bases complete the pending word
        const u64 hi = rest ? (bits >> (2 * rest)) : bits;
        const u64 word = room == 32 ? hi : ((acc << (2 * room)) | hi);
        if (rec_ok) flush_word(word);
        acc = rest ? (bits & (((u64)1 << (2 * rest)) - 1)) : 0;
        nacc = rest;
    };
    u64 w = 0;
    while (true) {
        if (MARK) visited[cur] = 1;
        w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
        if (FRAG && KC_SC_IS(w)) {
            // a contracted run: take its bases, continue at its far end
            if (w & KC_SC_NEXT) {
                // the run and the step out of it in one word: no look at the far end
                const u64 nb1 = ((w >> 45) & 7) + 1;
                if (REC) push_group(((((w >> 48) & 0x3fffull) >> (14 - 2 * (nb1 - 1))) << 2) | ((w >> 43) & 3), (int)nb1);
                len += nb1;
                cur = (w >> 6) & (((u64)1 << 36) - 1);
                co = (int)((w >> 42) & 1);
                if (len > limit) {
                    if (__hip_atomic_load(cursor + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicExch((ull *)(cursor + 7), 1ULL);
                    return;
                }
                continue;
            }
            u64 nb;
            if (w & KC_SC_EXT) {
                nb = (w >> 21) & 0x3fff;
                const u64 *src = frag.pool + (w >> 35);
                if (REC)
                    for (u64 i = 0; i < nb; i += 32) {
                        const int m = nb - i >= 32 ? 32 : (int)(nb - i);
                        push_group(src[i >> 5] >> (64 - 2 * m), m);
                    }
            } else {
                nb = (w >> 21) & 31;
                if (REC && nb) push_group(w >> (64 - 2 * nb), (int)nb);
            }
            len += nb;
            cur += (u64)(long long)((int)(((u32)(w >> 6) & 0x3fffu) ^ 0x2000u) - 0x2000);   // the far node: a signed 14-bit distance in ids
            co = (int)((w >> 20) & 1);
            if (len > limit) {
                if (__hip_atomic_load(cursor + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicExch((ull *)(cursor + 7), 1ULL);
                return;
            }
            w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
        }
        if (!LK_IS_MATCHED(w)) break;
        if (REC) push_group(LK_BASE(w), 1);
        cur = LK_SLOT(w);
        co = LK_PORT(w) == 0 ? 0 : 1;
        if (++len > limit) {
            if (__hip_atomic_load(cursor + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicExch((ull *)(cursor + 7), 1ULL);
            return;
        }
        // once any walk has hit the limit the whole pass is void: stop wasting steps (checked every 256 steps)
        if (limit != ~(u64)0 && (len & 255) == 0 && __hip_atomic_load(cursor + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    }
    // here w is the word of the port the walk ends on: dead, carrying the slot of the tail k-mer (as the head's entry port does)
    const u64 hslot = LK_SLOT(link[h]), tslot = LK_SLOT(w);
    Key<W> hk = set_key_at(set, hslot);
    bool winner = true, turned = false;
    if (complements) {
        const Key<W> tk = set_key_at(set, tslot);
        winner = (len == 1) ? (o == 0) : key_lt(hk, tk);
        const u64 far = (cur << 1) | (u64)(co == 0 ? 1 : 0);   // the dead port the walk ended on = the head at the other end
        if (link_marks && kc_walk_class(far) != (u32)cls_pass) {
            // (a later class: an earlier one would have marked this head) -- the only walk of this chain
            link_marks[far] = w | LK_WALKED;
            if (!winner) { turned = true; hk = tk; }
            winner = true;
        }
    }
    if (!winner) return;
    if (REC) {
        if (nacc && rec_ok) flush_word(acc << (64 - 2 * nacc));
        if (!rec_ok) atomicExch((ull *)(cursor + 6), 1ULL);   // pool exhausted (cannot happen with the caller's sizing): no stream emit
    }
    atomicAdd((ull *)(cursor + 3), (ull)len);   // nodes covered by chains that have ends
    u64 pos = atomicAdd((ull *)cursor, 1ULL);
    chain_head[pos] = h;
    chain_len[pos] = len;
    for (int i = 0; i < W; ++i) chain_key[pos * W + i] = hk.w[i];
    if (REC) { rec.chain_stream[pos] = (u64)first_chunk | (turned ? KC_STREAM_TURNED : 0); rec.chain_tail[pos] = (tslot << 1) | (u64)co; }   // (the tail by SLOT: the stream emit only needs its key)
}

template <bool FILL>
__global__ void __launch_bounds__(256) kc_unvisited_kernel(u64 n_nodes, const u32 *__restrict__ ncount, u32 nstride, const u8 *__restrict__ visited, u64 *list, u64 *cursor) {
    u64 nid = kc_gtid();
    if (nid >= n_nodes || visited[nid] || !kc_nid_in_use(nid, ncount, nstride)) return;
    u64 pos = atomicAdd((ull *)cursor, 1ULL);
    if (FILL) list[pos] = nid;
}

__global__ void __launch_bounds__(256) kc_iota_kernel(u64 *a, u64 n) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = t;
}
__global__ void __launch_bounds__(256) kc_gather_word_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ perm, u64 n, int W,
                                                             int word, u64 *out) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = keys[perm[t] * W + word];
}
// lens (bases) and word counts (len/32 + 1: always at least one padding base, so that packed simplitigs can be fed
// to the k-mer insert kernel with the padding marked as a break)
__global__ void __launch_bounds__(256) kc_chain_sizes_kernel(const u64 *__restrict__ chain_len, const u64 *__restrict__ perm, u64 n,
                                                             int k, u64 *lens, u64 *nwords) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 L = chain_len[perm[t]] + (u64)(k - 1);
    lens[t] = L;
    nwords[t] = L / 32 + 1;
}
__global__ void kc_finish_offsets_kernel(u64 *woff, const u64 *nwords, u64 n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) woff[n] = n ? woff[n - 1] + nwords[n - 1] : 0;
}

// emit pass: one thread per chain (in final order): spell the simplitig into packed 2-bit words (32 bases per word,
// first base in bits 63:62) and record its first and last k-mer.
template <int W>
__global__ void __launch_bounds__(256) kc_walk_emit_kernel(SetView<W> set, const u64 *__restrict__ link, int k,
                                                           const u64 *__restrict__ chain_head, const u64 *__restrict__ perm, u64 n,
                                                           const u64 *__restrict__ woff, u64 *words, u64 *firstK, u64 *lastK) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 h = chain_head[perm[t]];
    u64 cur = h >> 1;
    int co = (int)(h & 1);
    Key<W> kx = set_key_at(set, LK_SLOT(link[h]));   // (the head's entry port is dead: its word carries the slot)
    if (co) kx = key_rc(kx, k);
    for (int i = 0; i < W; ++i) firstK[t * W + i] = kx.w[i];
    u64 wp = woff[t], acc = 0;
    int nacc = 0;
    for (int i = 0; i < k; ++i) {
        acc = (acc << 2) | key_at(kx, k, i);
        if (++nacc == 32) { words[wp++] = acc; acc = 0; nacc = 0; }
    }
    while (true) {
        u64 w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
        if (!LK_IS_MATCHED(w)) break;
        u64 b = LK_BASE(w);
        kx = key_append(kx, b, k);
        acc = (acc << 2) | b;
        if (++nacc == 32) { words[wp++] = acc; acc = 0; nacc = 0; }
        cur = LK_SLOT(w);
        co = LK_PORT(w) == 0 ? 0 : 1;
    }
    words[wp] = nacc ? (acc << (64 - 2 * nacc)) : 0;
    for (int i = 0; i < W; ++i) lastK[t * W + i] = kx.w[i];
}

// emit from the recordings of the measure pass: first k bases from the head key, then the recorded appended bases.  EIGHT LANES PER
// SIMPLITIG: output word j holds the bases 32 j .. 32 j + 31 of the simplitig (the last word, always written, the remainder or
// nothing), so every lane can build its words on its own -- from the head key, two recorded words and a shift -- and the lanes of a group
// read and write consecutive words.  (One thread per simplitig, walking its recording word by word, had every lane of a wavefront in a
// line of its own for reads and writes alike: each 8-byte access moved a whole line between L2 and the CU -- 2.5 ms for the 0.6 GB of
// cfg[2].)  A turned chain (the recording runs from the chain's last k-mer to its first) is the reverse complement of (head key,
// recorded bases): its word j is the turned-round window that ends L - 32 j bases into that source.
template <int W, int LG>
__global__ void __launch_bounds__(256) kc_stream_emit_kernel(SetView<W> set, const u64 *__restrict__ link, int k, const u64 *__restrict__ chain_head,
                                                             const u64 *__restrict__ chain_len, const u64 *__restrict__ chain_stream,
                                                             const u64 *__restrict__ chain_tail, const u64 *__restrict__ perm, u64 n,
                                                             const u64 *__restrict__ pool, const u32 *__restrict__ chunk_next,
                                                             const u64 *__restrict__ woff, u64 *words, u64 *firstK, u64 *lastK) {
    const u64 gt = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr u32 G = 1u << LG;   // lanes per simplitig: 8 for reads of a kilobase, fewer for simplitigs of a few words
    const u64 t = gt >> LG;
    const u32 g = (u32)(gt & (G - 1));
    if (t >= n) return;
    const u64 p = perm[t];
    const u64 h = chain_head[p], len = chain_len[p];
    const bool turned = (chain_stream[p] & KC_STREAM_TURNED) != 0;
    Key<W> kx = set_key_at(set, LK_SLOT(link[h]));   // (the head's entry port is dead: its word carries the slot; the tail is listed by slot)
    if (h & 1) kx = key_rc(kx, k);
    if (g == 0) {
        const u64 tail = chain_tail[p];
        Key<W> kl = set_key_at(set, tail >> 1);
        if (tail & 1) kl = key_rc(kl, k);
        const Key<W> fk = turned ? key_rc(kl, k) : kx, lk = turned ? key_rc(kx, k) : kl;
        for (int i = 0; i < W; ++i) { firstK[t * W + i] = fk.w[i]; lastK[t * W + i] = lk.w[i]; }
    }
    const u64 L = (u64)k + len - 1, nw = (L >> 5) + 1, nrec = (len - 1 + 31) >> 5, w0 = woff[t];
    // recorded word r (none beyond the recording); asked for in ascending order only: the chunks are a singly linked list
    u32 chunk = (u32)chain_stream[p], ci = 0;
    auto rec = [&](u64 r) -> u64 {
        if (r >= nrec) return 0;
        const u32 c = (u32)(r / KC_CHUNK_WORDS);
        while (ci < c) { chunk = chunk_next[chunk]; ++ci; }
        return pool[(u64)chunk * KC_CHUNK_WORDS + (r % KC_CHUNK_WORDS)];
    };
    // the 32 bases of the source (head key, then the recording) from position i0 on, first one in the highest pair
    auto src32 = [&](u64 i0) -> u64 {
        u64 x = 0;
        int have = 0;
        if (i0 < (u64)k) {
            have = (u64)k - i0 >= 32 ? 32 : (int)((u64)k - i0);
            for (int j = 0; j < have; ++j) x = (x << 2) | key_at(kx, k, (int)i0 + j);
            if (have < 32) x <<= 64 - 2 * have;
        }
        if (have < 32) {
            const u64 o = i0 + (u64)have - (u64)k, r = o >> 5;
            const int sh = (int)(o & 31) * 2;
            const u64 a = rec(r);
            const u64 y = sh ? ((a << sh) | (rec(r + 1) >> (64 - sh))) : a;
            x |= have ? (y >> (2 * have)) : y;
        }
        return x;
    };
    if (!turned) {
        for (u64 j = g; j < nw; j += G) {
            const u64 valid = L - 32 * j >= 32 ? 32 : L - 32 * j;
            u64 x = valid ? src32(32 * j) : 0;
            if (valid && valid < 32) x &= ~(u64)0 << (64 - 2 * valid);
            words[w0 + j] = x;
        }
        return;
    }
    // turned: the words from the last one down, so that the source positions -- and the recorded words asked for -- ascend
    const u64 jtop = nw - 1;                                   // (word jtop holds the L & 31 last bases of the turned simplitig: the source's first ones)
    for (u64 q = g; q < nw; q += G) {
        const u64 j = jtop - q;
        const u64 valid = L - 32 * j >= 32 ? 32 : L - 32 * j;
        u64 x = 0;
        if (valid == 32) x = kc_rc_word(src32(L - 32 * j - 32));
        else if (valid) x = kc_rc_word(src32(0) & (~(u64)0 << (64 - 2 * valid))) << (2 * (32 - valid));
        words[w0 + j] = x;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Segment-parallel chain processing.  Starts = chain heads + oriented splitters whose entry port is matched; every start
// walks its segment (to the next splitter or the chain tail) in parallel; one thread per head then walks the short list
// of its segments (length/256 steps) to get the chain's length, tail and segment offsets; segments finally emit their
// bases in parallel.  A multi-Mbp chain therefore costs ~256 sequential steps instead of millions.
#define KC_SEG_NONE (~(u64)0)
#define KC_SEG_LOSER (~(u64)1)   /* seg_chain: the segment belongs to the direction of a chain that is not emitted (reached from a head all the same) */
__global__ void __launch_bounds__(256) kc_split_starts_kernel(const u64 *__restrict__ link, u64 n_nodes, const u32 *__restrict__ ncount, u32 nstride,
                                                              int complements, u64 *starts, u64 *cursor) {
    // splitters are the nodes whose dense id is a multiple of 256 (round 1 flags exactly those LK_SPLIT)
    const u64 slot = ((u64)blockIdx.x * blockDim.x + threadIdx.x) << 8;
    if (slot >= n_nodes || !kc_nid_in_use(slot, ncount, nstride)) return;
    const int no = complements ? 2 : 1;
    for (int o = 0; o < no; ++o) {
        u64 w = link[(slot << 1) | (u64)o];          // entry port of orientation o is port o
        if (!LK_IS_MATCHED(w)) continue;              // dead entry port: this oriented node is a head (already listed)
        u64 pos = atomicAdd((ull *)cursor, 1ULL);
        starts[pos] = (slot << 1) | (u64)o;
    }
}
__device__ __forceinline__ u64 kc_find_start(const u64 *starts, u64 n, u64 id) {
    u64 lo = 0, hi = n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (starts[mid] < id) lo = mid + 1; else hi = mid; }
    return lo;   // callers only look up ids that are present
}
// one thread per start (sorted ascending): nodes until the next splitter (exclusive) or the tail (inclusive)
// MODE 0: measure (seg_len / seg_next / seg_tail).  MODE 1: measure and mark in `visited` -- the k-mers on cycles WITHOUT a
// splitter are then the ones no mark accounts for, and no marking walk of its own (as long as this one: a scattered line per node)
// is needed.  MODE 2: the segments that kc_chain_reduce_kernel assigned to NO chain (seg_chain: their splitters lie on cycles, no
// head reaches them) take their marks back; nothing else is written.
// nstride != 0: SPARSE marks -- only the nodes where the walk enters a block (reached by a link from another block; the start node
// when it is a head or entered from another block; the tail): one scattered byte per in-block run instead of one per node
// (marking every node made this walk three times as slow); kc_frag_unvisited_kernel reads them.
template <int MODE>
__global__ void __launch_bounds__(256) kc_segment_walk_kernel(const u64 *__restrict__ link, const u64 *__restrict__ starts, u64 nstarts,
                                                              u64 *seg_len, u64 *seg_next, u64 *seg_tail, u8 *visited,
                                                              const u64 *__restrict__ seg_chain, u64 max_len, u32 nstride) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nstarts) return;
    if (MODE == 2 && seg_chain[j] != KC_SEG_NONE) return;
    u64 cur = starts[j] >> 1;
    int co = (int)(starts[j] & 1);
    u64 len = 1;
    u64 w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
    u64 lo = 0;          // sparse marks: first id of the block the walk is in
    bool entered = true; // ... and whether `cur` was entered from outside it
    if (MODE != 0 && nstride) {
        lo = cur / nstride * nstride;
        const u64 win = link[(cur << 1) | (u64)co];   // (the entry port's word: same 16 bytes)
        entered = !LK_IS_MATCHED(win) || LK_SLOT(win) - lo >= (u64)nstride;
    }
    while (true) {
        const bool tail = !LK_IS_MATCHED(w);
        if (MODE != 0 && (entered || tail)) visited[cur] = MODE == 1 ? 1 : 0;
        if (tail) { if (MODE != 2) { seg_next[j] = KC_SEG_NONE; seg_tail[j] = (LK_SLOT(w) << 1) | (u64)co; } break; }   // (the tail by SLOT, from its dead port's word)
        u64 nxt = LK_SLOT(w);
        int no = LK_PORT(w) == 0 ? 0 : 1;
        u64 wn = link[(nxt << 1) | (u64)(no == 0 ? 1 : 0)];
        if (wn & LK_SPLIT) { if (MODE != 2) { seg_next[j] = kc_find_start(starts, nstarts, (nxt << 1) | (u64)no); seg_tail[j] = KC_SEG_NONE; } break; }
        if (MODE != 0 && nstride) {
            entered = nxt - lo >= (u64)nstride;
            if (entered) lo = nxt / nstride * nstride;
        }
        cur = nxt; co = no; w = wn;
        if (++len > max_len) { if (MODE != 2) { seg_next[j] = KC_SEG_NONE; seg_tail[j] = 0; } break; }   // (cannot happen on consistent links: a segment ends at a splitter or a chain end; never spin)
    }
    if (MODE != 2) seg_len[j] = len;
}
// one thread per start that is a chain head: walk the list of segments
template <int W>
__global__ void __launch_bounds__(256) kc_chain_reduce_kernel(SetView<W> set, const u64 *__restrict__ link, int complements,
                                                              const u64 *__restrict__ starts, u64 nstarts, const u64 *__restrict__ seg_len,
                                                              const u64 *__restrict__ seg_next, const u64 *__restrict__ seg_tail,
                                                              u64 *seg_chain, u64 *seg_off, u64 *chain_head, u64 *chain_len, u64 *chain_key,
                                                              u64 *cursor) {
    u64 j0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j0 >= nstarts) return;
    const u64 id = starts[j0];
    if (LK_IS_MATCHED(link[id]) || KC_SC_IS(link[id])) return;   // entry port matched (or the inward port of a contracted run): a splitter start, not a head
    u64 total = 0, j = j0, tail = 0;
    while (true) {
        total += seg_len[j];
        if (seg_next[j] == KC_SEG_NONE) { tail = seg_tail[j]; break; }
        j = seg_next[j];
    }
    Key<W> hk = set_key_at(set, LK_SLOT(link[id]));   // (a head's entry port is dead: its word carries the slot)
    bool winner = true;
    if (complements) winner = (total == 1) ? ((id & 1) == 0) : key_lt(hk, set_key_at(set, tail >> 1));
    if (!winner) {   // (its segments are told apart from those that no head reaches: the splitters on cycles)
        j = j0;
        while (true) {
            seg_chain[j] = KC_SEG_LOSER;
            if (seg_next[j] == KC_SEG_NONE) break;
            j = seg_next[j];
        }
        return;
    }
    atomicAdd((ull *)(cursor + 3), (ull)total);
    u64 pos = atomicAdd((ull *)cursor, 1ULL);
    chain_head[pos] = id;
    chain_len[pos] = total;
    for (int i = 0; i < W; ++i) chain_key[pos * W + i] = hk.w[i];
    u64 off = 0;
    j = j0;
    while (true) {
        seg_chain[j] = pos;
        seg_off[j] = off;
        off += seg_len[j];
        if (seg_next[j] == KC_SEG_NONE) break;
        j = seg_next[j];
    }
}
__global__ void __launch_bounds__(256) kc_inverse_perm_kernel(const u64 *__restrict__ perm, u64 n, u64 *rank) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) rank[perm[t]] = t;
}
// one thread per segment of an emitted chain: spell its bases into the (zeroed) packed words with 64-bit atomic ORs
template <int W>
__global__ void __launch_bounds__(256) kc_segment_emit_kernel(SetView<W> set, const u64 *__restrict__ link, int k,
                                                              const u64 *__restrict__ starts, u64 nstarts, const u64 *__restrict__ seg_chain,
                                                              const u64 *__restrict__ seg_off, const u64 *__restrict__ rank,
                                                              const u64 *__restrict__ woff, u64 *words, u64 *firstK, u64 *lastK) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nstarts) return;
    const u64 ch = seg_chain[j];
    if (ch >= KC_SEG_LOSER) return;   // (KC_SEG_NONE / KC_SEG_LOSER: not a segment of an emitted direction)
    const u64 r = rank[ch], wb = woff[r], off = seg_off[j];
    u64 cur = starts[j] >> 1;
    int co = (int)(starts[j] & 1);
    // the start's k-mer: a head's slot is in the word of its (dead) entry port, a splitter's is searched for (one per 256 k-mers)
    const u64 w_in = link[starts[j]];
    Key<W> kx = set_key_at(set, LK_IS_MATCHED(w_in) ? set_slot_of_nid(set, cur) : LK_SLOT(w_in));
    if (co) kx = key_rc(kx, k);
    u64 pos, acc = 0, wi;   // pos = next base position in the chain's string; acc = bits gathered for word wi
    auto put = [&](u64 b) {
        acc |= b << (62 - 2 * (pos & 31));
        ++pos;
        if ((pos & 31) == 0) { atomicOr((ull *)&words[wi], (ull)acc); acc = 0; ++wi; }
    };
    if (off == 0) {               // chain head: the whole first k-mer
        for (int i = 0; i < W; ++i) firstK[r * W + i] = kx.w[i];
        pos = 0; wi = wb;
        for (int i = 0; i < k; ++i) put((u64)key_at(kx, k, i));
    } else {                      // later segment: its first node contributes the last base of its oriented k-mer
        pos = (u64)(k - 1) + off; wi = wb + (pos >> 5);
        put(kx.w[0] & 3);
    }
    u64 w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
    while (true) {
        if (!LK_IS_MATCHED(w)) {  // tail of the chain
            for (int i = 0; i < W; ++i) lastK[r * W + i] = kx.w[i];
            break;
        }
        u64 nxt = LK_SLOT(w);
        int no = LK_PORT(w) == 0 ? 0 : 1;
        u64 wn = link[(nxt << 1) | (u64)(no == 0 ? 1 : 0)];
        if (wn & LK_SPLIT) break;
        u64 b = LK_BASE(w);
        kx = key_append(kx, b, k);
        put(b);
        cur = nxt; co = no; w = wn;
    }
    if (acc) atomicOr((ull *)&words[wi], (ull)acc);
}

// ---- the segment-parallel path over CONTRACTED links (kc_frag_contract_kernel with no_next): the walks hop from run to run, two
// scattered lines per in-block run (~9 k-mers) instead of one per k-mer.  A hop walk only ever lands on the ends of contracted runs
// and on the nodes of uncontracted stretches, so the splitters are chosen among the former: a node whose id is a multiple of 64
// and one of whose words is a shortcut (~1 in 290 k-mers).  Both the lister and a walk that lands on a node see that in the node's
// two words.  Marks (MODE 1 / 2) are the sparse ones of kc_segment_walk_kernel: they are read after the links have been restored.
#define KC_FSPLIT_MASK 63ull
__device__ __forceinline__ bool kc_hop_splitter(u64 nid, const ulonglong2 &ww) { return (nid & KC_FSPLIT_MASK) == 0 && (KC_SC_IS(ww.x) || KC_SC_IS(ww.y)); }
__device__ __forceinline__ ulonglong2 kc_link_pair(const u64 *__restrict__ link, u64 nid) { return *reinterpret_cast<const ulonglong2 *>(link + (nid << 1)); }
// the far end of the run a shortcut word stands for: node, orientation there, bases (= nodes - 1)
__device__ __forceinline__ void kc_hop_decode(u64 w, u64 cur, u64 &far, int &fo, u64 &nb) {
    nb = (w & KC_SC_EXT) ? ((w >> 21) & 0x3fffull) : ((w >> 21) & 31ull);
    far = cur + (u64)(long long)((int)(((u32)(w >> 6) & 0x3fffu) ^ 0x2000u) - 0x2000);
    fo = (int)((w >> 20) & 1);
}
__global__ void __launch_bounds__(256) kc_split_starts_hop_kernel(const u64 *__restrict__ link, u64 n_nodes, const u32 *__restrict__ ncount, u32 nstride,
                                                                  int complements, u64 *starts, u64 *cursor) {
    // (one reservation per workgroup: 1.8 10^7 adds on one word took 19 ms)
    const u64 nid = kc_gtid() << 6;
    u32 m = 0;   // bit o: the oriented node (nid, o) is a start
    if (nid < n_nodes && kc_nid_in_use(nid, ncount, nstride)) {
        const ulonglong2 ww = kc_link_pair(link, nid);
        if (kc_hop_splitter(nid, ww)) {
            const int no = complements ? 2 : 1;
            for (int o = 0; o < no; ++o) {
                const u64 e = o ? ww.y : ww.x;                           // entry port of orientation o is port o
                if (KC_SC_IS(e) || LK_IS_MATCHED(e)) m |= 1u << o;      // (a dead entry port: a head, already listed)
            }
        }
    }
    u64 pos = kc_block_reserve((u32)__popc(m), cursor);
    if (m & 1u) starts[pos++] = nid << 1;
    if (m & 2u) starts[pos] = (nid << 1) | 1ull;
}
template <int MODE>
__global__ void __launch_bounds__(256) kc_segment_hop_kernel(const u64 *__restrict__ link, const u64 *__restrict__ starts, u64 nstarts,
                                                             u64 *seg_len, u64 *seg_next, u64 *seg_tail, u8 *visited,
                                                             const u64 *__restrict__ seg_chain, u64 max_len, u32 nstride, int both_ends) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nstarts) return;
    if (MODE == 2 && seg_chain[j] != KC_SEG_NONE) return;
    u64 cur = starts[j] >> 1;
    int co = (int)(starts[j] & 1);
    ulonglong2 ww = kc_link_pair(link, cur);
    u64 len = 1, lo = 0;
    bool entered = false;   // `cur` was reached by a link from another block (or is a chain's head)
    if (MODE != 0) {
        lo = cur / nstride * nstride;
        const u64 win = co == 0 ? ww.x : ww.y;
        entered = !KC_SC_IS(win) && (!LK_IS_MATCHED(win) || LK_SLOT(win) - lo >= (u64)nstride);
    }
    u64 next = KC_SEG_NONE, tailv = KC_SEG_NONE;
    while (true) {
        const u64 wl = co == 0 ? ww.y : ww.x;   // the word of the port the walk leaves through
        if (KC_SC_IS(wl)) {
            u64 far, nb; int fo;
            kc_hop_decode(wl, cur, far, fo, nb);
            // (bidirectional sets: a run is entered at both ends, by the walks of the two directions; kc_frag_unvisited_kernel looks at
            // the mark of its SMALLER end only -- half the scattered stores)
            if (MODE != 0 && entered && (!both_ends || far > cur)) visited[cur] = MODE == 1 ? 1 : 0;
            const ulonglong2 wf = kc_link_pair(link, far);
            if (kc_hop_splitter(far, wf)) { len += nb - 1; next = (far << 1) | (u64)fo; break; }
            len += nb; cur = far; co = fo; ww = wf; entered = false;
            if (len > max_len) { tailv = 0; break; }   // (cannot happen on consistent links; never spin)
            continue;
        }
        const bool tail = !LK_IS_MATCHED(wl);
        if (MODE != 0 && (entered || tail)) visited[cur] = MODE == 1 ? 1 : 0;
        if (tail) { tailv = (LK_SLOT(wl) << 1) | (u64)co; break; }   // (the tail by SLOT, from its dead port's word)
        const u64 nxt = LK_SLOT(wl);
        const int no = LK_PORT(wl) == 0 ? 0 : 1;
        const ulonglong2 wn = kc_link_pair(link, nxt);
        if (kc_hop_splitter(nxt, wn)) { next = (nxt << 1) | (u64)no; break; }
        if (MODE != 0) { entered = nxt - lo >= (u64)nstride; if (entered) lo = nxt / nstride * nstride; }
        cur = nxt; co = no; ww = wn;
        if (++len > max_len) { tailv = 0; break; }
    }
    if (MODE != 2) {
        seg_len[j] = len;
        seg_next[j] = next == KC_SEG_NONE ? KC_SEG_NONE : kc_find_start(starts, nstarts, next);
        seg_tail[j] = tailv;
    }
}
// one thread per segment of an emitted chain, over contracted links: the bases of a run leave as groups
template <int W>
__global__ void __launch_bounds__(256) kc_segment_hop_emit_kernel(SetView<W> set, const u64 *__restrict__ link, int k, const u64 *__restrict__ fpool,
                                                                  const u64 *__restrict__ starts, u64 nstarts, const u64 *__restrict__ seg_chain,
                                                                  const u64 *__restrict__ seg_off, const u64 *__restrict__ rank,
                                                                  const u64 *__restrict__ woff, u64 *words, u64 *firstK, u64 *lastK) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nstarts) return;
    const u64 ch = seg_chain[j];
    if (ch >= KC_SEG_LOSER) return;
    const u64 r = rank[ch], wb = woff[r], off = seg_off[j];
    u64 cur = starts[j] >> 1;
    int co = (int)(starts[j] & 1);
    ulonglong2 ww = kc_link_pair(link, cur);
    u64 pos, acc = 0, wi;   // pos = next base position in the chain's string; acc = bits gathered for word wi
    // n (1..32) bases, right-aligned in `bits`, first base in the highest pair
    auto put_group = [&](u64 bits, int n) {
        const int room = 32 - (int)(pos & 31);
        pos += (u64)n;
        if (n < room) { acc |= bits << (2 * (room - n)); return; }
        const int rest = n - room;
        acc |= rest ? (bits >> (2 * rest)) : bits;
        atomicOr((ull *)&words[wi], (ull)acc);
        ++wi;
        acc = rest ? bits << (64 - 2 * rest) : 0;
    };
    {
        // the start's k-mer: a head's slot is in the word of its (dead) entry port, a splitter's is searched for
        const u64 w_in = co == 0 ? ww.x : ww.y;
        Key<W> kx = set_key_at(set, LK_STATE(w_in) == LK_DEAD ? LK_SLOT(w_in) : set_slot_of_nid(set, cur));
        if (co) kx = key_rc(kx, k);
        if (off == 0) {               // chain head: the whole first k-mer
            for (int i = 0; i < W; ++i) firstK[r * W + i] = kx.w[i];
            pos = 0; wi = wb;
            for (int i = 0; i < k; ++i) put_group((u64)key_at(kx, k, i), 1);
        } else {                      // later segment: its first node contributes the last base of its oriented k-mer
            pos = (u64)(k - 1) + off; wi = wb + (pos >> 5);
            put_group(kx.w[0] & 3, 1);
        }
    }
    while (true) {
        const u64 wl = co == 0 ? ww.y : ww.x;
        if (KC_SC_IS(wl)) {
            u64 far, nb; int fo;
            kc_hop_decode(wl, cur, far, fo, nb);
            const ulonglong2 wf = kc_link_pair(link, far);
            const bool stop = kc_hop_splitter(far, wf);   // (the step onto a splitter belongs to the splitter's segment)
            const u64 total = nb - (stop ? 1 : 0);
            if (wl & KC_SC_EXT) {
                const u64 *src = fpool + (wl >> 35);
                for (u64 i = 0; i < total; i += 32) {
                    const int m = total - i >= 32 ? 32 : (int)(total - i);
                    put_group(src[i >> 5] >> (64 - 2 * m), m);
                }
            } else if (total) put_group((wl >> (64 - 2 * nb)) >> (2 * (nb - total)), (int)total);
            if (stop) break;
            cur = far; co = fo; ww = wf;
            continue;
        }
        if (!LK_IS_MATCHED(wl)) {     // tail of the chain: its k-mer by the slot in the dead word
            Key<W> kt = set_key_at(set, LK_SLOT(wl));
            if (co) kt = key_rc(kt, k);
            for (int i = 0; i < W; ++i) lastK[r * W + i] = kt.w[i];
            break;
        }
        const u64 nxt = LK_SLOT(wl);
        const int no = LK_PORT(wl) == 0 ? 0 : 1;
        const ulonglong2 wn = kc_link_pair(link, nxt);
        if (kc_hop_splitter(nxt, wn)) break;
        put_group(LK_BASE(wl), 1);
        cur = nxt; co = no; ww = wn;
    }
    if (acc) atomicOr((ull *)&words[wi], (ull)acc);
}

// ---------------------------------------------------------------------------------------------------------
static int read_u64(kc_ctx *ctx, const u64 *d, u64 *h);
int kcp_sort_perm_by_keys(kc_ctx *ctx, const u64 *d_keys, u64 n, int W, int k, u64 **perm_out) {   // (also the sharded flow's simplitig order: kc_shard.hip)
    // LSD over the key words (least significant word first), each pass a stable radix sort of (word, perm).
    u64 *perm = nullptr, *perm2 = nullptr, *kw = nullptr, *kw2 = nullptr;
    KcScope scope(ctx);
    KC_TRY(scope.alloc(&perm, n));
    KC_TRY(scope.alloc(&perm2, n));
    KC_TRY(scope.alloc(&kw, n));
    KC_TRY(scope.alloc(&kw2, n));
    KC_LAUNCH(kc_iota_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, perm, n);
    // many keys (read sets: one chain per read): one MSD bucket pass + LDS sorts (kcu_sort_rows_bucketed); the LSD passes
    // below remain for small inputs and for key sets so skewed that a bucket overflows
    if (n >= 1024 && !getenv("KC_NO_BUCKET_SORT")) {
        u64 *kout = nullptr, *flag = nullptr;
        KC_TRY(scope.alloc(&kout, n * (u64)W));
        KC_TRY(scope.alloc(&flag, 1));
        const int brc = kcu_sort_rows_bucketed(ctx, d_keys, perm, n, W, 2 * k, kout, perm2, flag);
        if (brc == KC_OK) {
            u64 ovf = 1;
            KC_TRY(read_u64(ctx, flag, &ovf));
            if (!ovf) { scope.keep(perm2); *perm_out = perm2; return KC_OK; }
        } else if (brc != KC_E_ARG) return brc;
        scope.release(kout);
    }
    for (int word = 0; word < W; ++word) {
        int bits = 2 * k - 64 * word;
        if (bits <= 0) break;
        if (bits > 64) bits = 64;
        KC_LAUNCH(kc_gather_word_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, d_keys, perm, n, W, word, kw);
        KC_TRY(kcu_sort_pairs(ctx, kw, kw2, perm, perm2, n, 0, bits));
        std::swap(perm, perm2);
    }
    KC_HIP(ctx, hipGetLastError());
    scope.keep(perm);   // (perm and perm2 may have been swapped: keep whichever holds the result)
    *perm_out = perm;
    return KC_OK;
}

static int read_u64(kc_ctx *ctx, const u64 *d, u64 *h) {
    KC_HIP(ctx, hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}

template <int W> static int build_simplitigs_w(kc_set *set, kc_simplitigs **out, u64 *rounds_out, u64 *cycles_out) {
    u64 cycles_cut = 0;
    kc_ctx *ctx = set->ctx;
    KcScope scope(ctx);
    const int k = set->k, compl_ = set->complements;
    SetView<W> sv = kc_view<W>(set);
    const u64 cap = set->cap;
    const int xcd_aware = getenv("KC_NOXCD") ? 0 : 1;
    KC_TRY(kcs_refresh_size(set));
    const u64 N = set->size_host;
    if (N == 0) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
    // Nodes are DENSE ids: the k-mers numbered in slot order (kc_set_dev.h).  The port links are 16 bytes per k-mer, written
    // contiguously by round 1, not 16 bytes per slot of a table at load 0.5.
    KC_TRY(kcs_prepare_dense(set));
    sv = kc_view<W>(set);
    const u64 NID = set->nblocks * (u64)set->nstride;   // node ids (block b owns [b * nstride, b * nstride + its k-mers))
    const u32 *ncount = set->ncount;
    const u32 nstride = set->nstride;
    const u64 nports = NID * 2;
    u64 *link = nullptr, *cnt = nullptr;
    KC_TRY(scope.alloc(&link, nports + 2));
    KC_TRY(scope.alloc(&cnt, 8));
    // (no clearing of `link`: round 1 writes both words of every k-mer)
    KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
    // heads (ports that die) and the work list of unresolved ports are collected on the fly into bounded buffers; an
    // overflow (never seen on read-like inputs) falls back to exact counting passes.
    // (sized for read-like inputs -- 2 heads per simplitig, hardly any branch port; a cold process pays ~26 ms per GB it obtains
    // from hipMalloc, so generous bounds are not free.  Overflows take the exact paths below.)
    u64 heads_cap = N / 32 + 65536, work_cap = N / 64 + 65536;
    // (reads with coverage -- the build's sample saw every k-mer several times -- have an error-made tip or bubble every few dozen
    // k-mers: 19.7 10^6 simplitigs and as many branch ports per 7.3 10^8 k-mers at 30x with 1 % errors; the exact paths cost 80 ms there)
    if (set->occ_ratio >= 1.5f) { heads_cap = N / 6 + 65536; work_cap = N / 6 + 65536; }
    u64 *heads = nullptr, *work_a = nullptr, *work_b = nullptr;
    KC_TRY(scope.alloc(&heads, heads_cap));
    KC_TRY(scope.alloc(&work_a, work_cap));
    KC_TRY(scope.alloc(&work_b, work_cap));
    u64 *heads_cursor = cnt + 2, *work_cursor = cnt;
    // round 1 over all k-mers (the dominant kernel of this phase: 8 table probes per k-mer)
    u64 *branch = work_b, *branch_cursor = cnt + 4;   // branch ports of round 1 (work_b is free until round 2)
    // sharded round-1 lists (gathered into `heads` / `branch` right after the kernel)
    u64 *heads_sh = nullptr, *branch_sh = nullptr, *shard_cur = nullptr;
    const u64 heads_shard_cap = heads_cap / KC_NSHARD, branch_shard_cap = work_cap / KC_NSHARD;
    KC_TRY(scope.alloc(&heads_sh, heads_shard_cap * KC_NSHARD));
    KC_TRY(scope.alloc(&branch_sh, branch_shard_cap * KC_NSHARD));
    KC_TRY(scope.alloc(&shard_cur, 2 * KC_NSHARD * KC_SHARD_STRIDE));
    KC_HIP(ctx, hipMemsetAsync(shard_cur, 0, 2 * KC_NSHARD * KC_SHARD_STRIDE * 8, ctx->stream));
    u64 *heads_cursors = shard_cur, *branch_cursors = shard_cur + KC_NSHARD * KC_SHARD_STRIDE;
    KC_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    // compile-time k for the benchmark configurations (k = 31 / 63 / 127 with m = 15): constant trip counts everywhere
    constexpr int KSPEC = W == 1 ? 31 : (W == 2 ? 63 : 127);
    const bool kspec = k == KSPEC && set->mlen == KC_MSPEC && !getenv("KC_NO_KSPEC");
    u64 *spill = nullptr, *spill_cursor = cnt + 6;
    // ports the block kernel hands on to the generic lookups: ~0.1 % of the ports of a table at load 0.5, 1.6 % (3.2 % of N) at the load
    // 0.66 that sets beyond ~3.2 10^9 k-mers are built with to stay inside the device memory -- measured at 4.4 10^9 k-mers: 1.41 10^8.  A list
    // that overflows means round 1 again by the generic kernel (1.4 s there against 0.17 s for the block kernel): sized for the load at hand.
    const u64 spill_cap = (cap >= N + N / 2 + N / 3 ? N / 64 : N / 12) + 65536;
    if (set->mlen) {
        KC_TRY(scope.alloc(&spill, spill_cap));
        // middle-based homes: one workgroup per block, tags (and single-word keys of blocks up to 8192 slots) staged in LDS
        const bool stage_keys = W == 1 && set->bshift <= 13 && !getenv("KC_NOSTAGE");
        const int NT1 = getenv("KC_PROBE_NT512") ? 512 : 1024;   // (tuning knob: the 512-thread form of the single-word kernel)
        const int nt = (stage_keys && NT1 == 1024) ? 1024 : 512;
        const bool mm = nt == 512;
        const size_t lds = kc_probe_lds_bytes(set->bshift, W, stage_keys, nt, mm) + (getenv("KC_PROBE_PAD") ? (size_t)atoi(getenv("KC_PROBE_PAD")) : 0);
        const unsigned pgrid = (unsigned)((set->nblocks + 7) / 8 * 8);
        auto launch = [&](auto kern) -> int {
            KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (getenv("KC_SIMP_TRACE")) {
                int nb = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), nt, lds);
                fprintf(stderr, "[simp] probe kernel: %d threads, %zu bytes of LDS per workgroup, %d workgroups per CU\n", nt, lds, nb);
            }
            KC_LAUNCH(kern, dim3(pgrid), dim3(nt), lds, ctx->stream, sv, link, k, compl_, xcd_aware | (getenv("KC_PROBE_LISTED_GLOBAL") ? 2 : 0) | (getenv("KC_PROBE_NO_SHARE") ? 4 : 0) | (KC_EXP_KNOB("KC_PROBE_DOWN_SPILL") ? 16 : 0), heads_sh, heads_shard_cap,
                               heads_cursors, branch_sh, branch_shard_cap, branch_cursors, spill, spill_cap, spill_cursor, ShLists{});
            return KC_OK;
        };
        // GP: every key sits at its primary home (a fresh bucketed build without spills): the sweep guesses "the candidates live in my
        // own block" and computes no minimizers (see the kernel); KC_PROBE_NO_GUESS=1: the form that computes them for every k-mer
        const bool gp = set->homes_primary && !getenv("KC_PROBE_NO_GUESS");
        if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] probe kernel: in-block guess %s\n", gp ? "on" : "off");
        if (stage_keys) {
            if constexpr (W == 1) {
                if (nt == 1024) {
                    if (gp) KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, true, 1024, false, false, true>) : launch(kc_probe_block_kernel<W, 0, true, 1024, false, false, true>));
                    else KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, true, 1024, false>) : launch(kc_probe_block_kernel<W, 0, true, 1024, false>));
                } else KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, true, 512, true>) : launch(kc_probe_block_kernel<W, 0, true, 512, true>));
            }
        } else if (gp) KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, false, 512, true, false, true>) : launch(kc_probe_block_kernel<W, 0, false, 512, true, false, true>));
        else KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, false, 512, true>) : launch(kc_probe_block_kernel<W, 0, false, 512, true>));
    } else {
        // plain hashing (k < 18): per-key homes, 8 lookups per k-mer
        const unsigned cgrid = kc_grid8(cap, KC_CHOOSE_SLOTS);
        KC_LAUNCH((kc_choose_dense_kernel<W, 0>), dim3(cgrid), dim3(256), 0, ctx->stream, sv, link, k, compl_, xcd_aware,
                           heads_sh, heads_shard_cap, heads_cursors, branch_sh, branch_shard_cap, branch_cursors, 0);
    }
    KC_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    if (set->mlen) {
        // the ports the block kernel handed on (window search not applicable, external list full); if even their list overflowed,
        // round 1 is redone by generic lookups
        KC_LAUNCH((kc_probe_spill_kernel<W>), dim3(4096), dim3(256), 0, ctx->stream, sv, link, k, compl_, (const u64 *)spill, spill_cap,   // (grid-stride; 16 workgroups per CU: the lookups are dependent scattered reads)
                           (const u64 *)spill_cursor, heads_sh, heads_shard_cap, heads_cursors, branch_sh, branch_shard_cap, branch_cursors);
        u64 nspill = 0;
        KC_TRY(read_u64(ctx, spill_cursor, &nspill));
        if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] round 1: %llu ports handed to the generic lookups\n", (unsigned long long)nspill);
        if (nspill > spill_cap) {
            KC_HIP(ctx, hipMemsetAsync(shard_cur, 0, 2 * KC_NSHARD * KC_SHARD_STRIDE * 8, ctx->stream));
            KC_LAUNCH((kc_probe_generic_kernel<W>), kc_grid2(cap, 256), dim3(256), 0, ctx->stream, sv, link, k, compl_, heads_sh, heads_shard_cap,
                               heads_cursors, branch_sh, branch_shard_cap, branch_cursors);
        }
        scope.release(spill);
    }
    KC_LAUNCH(kc_gather_shards_kernel, dim3(8, KC_NSHARD), dim3(256), 0, ctx->stream, (const u64 *)heads_sh, heads_shard_cap,
                       (const u64 *)heads_cursors, heads, heads_cap, heads_cursor);
    KC_LAUNCH(kc_gather_shards_kernel, dim3(8, KC_NSHARD), dim3(256), 0, ctx->stream, (const u64 *)branch_sh, branch_shard_cap,
                       (const u64 *)branch_cursors, branch, work_cap, branch_cursor);
    u64 nbranch = 0, nwork = 0, rounds = 1;
    KC_TRY(read_u64(ctx, branch_cursor, &nbranch));
    if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] round 1: %llu branch ports (list of %llu)\n", (unsigned long long)nbranch, (unsigned long long)work_cap);
    scope.release(heads_sh); scope.release(branch_sh); scope.release(shard_cur);
    if (nbranch && nbranch <= work_cap) {
        KC_LAUNCH((kc_reject_kernel<W>), dim3(kc_grid(nbranch, 256)), dim3(256), 0, ctx->stream, sv, link, k, compl_, (const u64 *)branch,
                           nbranch, work_a, work_cap, work_cursor);
        KC_TRY(read_u64(ctx, work_cursor, &nwork));
    }
    if (nbranch > work_cap || nwork > work_cap) {
        // heavily branching graph: exact scan of all round-1 choices (count, then mark + list); rejections a partial
        // kc_reject_kernel run may already have set are simply found again
        KC_HIP(ctx, hipMemsetAsync(work_cursor, 0, 8, ctx->stream));
        KC_LAUNCH((kc_mutual_scan_kernel<false>), kc_grid2((NID + 31) / 32, 256), dim3(256), 0, ctx->stream, link, NID, ncount, nstride,
                           (u64 *)nullptr, work_cursor);
        KC_TRY(read_u64(ctx, work_cursor, &nwork));
        scope.release(work_a); scope.release(work_b);
        work_cap = nwork + 1;
        KC_TRY(scope.alloc(&work_a, work_cap));
        KC_TRY(scope.alloc(&work_b, work_cap));
        KC_HIP(ctx, hipMemsetAsync(work_cursor, 0, 8, ctx->stream));
        KC_LAUNCH((kc_mutual_scan_kernel<true>), kc_grid2((NID + 31) / 32, 256), dim3(256), 0, ctx->stream, link, NID, ncount, nstride, work_a,
                           work_cursor);
        KC_TRY(read_u64(ctx, work_cursor, &nwork));
    }
    while (nwork) {
        ++rounds;
        if (rounds > 100000) return kc_fail(ctx, KC_E_NO_DEVICE, "simplitig matching did not converge");
        KC_LAUNCH((kc_choose_kernel<W, false, 0>), dim3(kc_grid(nwork, 256)), dim3(256), 0, ctx->stream, sv, link, k, compl_,
                           (const u64 *)work_a, nwork, 0, heads, heads_cap, heads_cursor);
        KC_HIP(ctx, hipMemsetAsync(work_cursor, 0, 8, ctx->stream));
        KC_LAUNCH(kc_accept_kernel, dim3(kc_grid(nwork, 256)), dim3(256), 0, ctx->stream, link, (const u64 *)work_a,
                           nwork, work_b, work_cap, work_cursor);
        KC_TRY(read_u64(ctx, work_cursor, &nwork));   // shrinks every round, so work_cap still bounds it
        std::swap(work_a, work_b);
    }
    KC_HIP(ctx, hipGetLastError());
    scope.release(work_a); scope.release(work_b);
    if (rounds_out) *rounds_out = rounds;

    // chain heads
    u64 nheads = 0;
    const u64 nsplit_max = (NID + 255) >> 8;   // splitters: nodes with id % 256 == 0
    KC_TRY(read_u64(ctx, heads_cursor, &nheads));
    if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] %llu chain heads (list of %llu), %llu matching round(s)\n", (unsigned long long)nheads, (unsigned long long)heads_cap, (unsigned long long)rounds);
    if (nheads > heads_cap) {   // overflow: exact count + fill
        scope.release(heads);
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
        KC_LAUNCH((kc_heads_kernel<false>), kc_grid2((NID + 31) / 32, 256), dim3(256), 0, ctx->stream, (const u64 *)link, NID, ncount, nstride, compl_,
                           (u64 *)nullptr, cnt);
        KC_TRY(read_u64(ctx, cnt, &nheads));
        KC_TRY(scope.alloc(&heads, nheads + 1));
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
        KC_LAUNCH((kc_heads_kernel<true>), kc_grid2((NID + 31) / 32, 256), dim3(256), 0, ctx->stream, (const u64 *)link, NID, ncount, nstride, compl_, heads,
                           cnt);
    }
    // ---- chain discovery.  Pass 0 works on the heads round 1..n produced.  K-mers on cycles (every port matched) are
    // reached from no head: if the pass does not cover all N nodes, the cycles are cut (at their minimum key, on port L),
    // the cut ends join the head list, and the discovery runs once more -- so long chains keep the segment-parallel path
    // (and short ones the recorded-stream emit) whether or not the input has cycles.
    u64 KC_CHAIN_LIMIT = 4096;   // longer chains take the segment-parallel path
    if (const char *e = getenv("KC_CHAIN_LIMIT")) { long v = atol(e); if (v >= 1) KC_CHAIN_LIMIT = (u64)v; }   // tests force the segment path
    const bool want_rec = !getenv("KC_NOREC");
    const KcRec no_rec{};
    u64 *chain_head = nullptr, *chain_len = nullptr, *chain_key = nullptr;
    u64 *starts = nullptr, *seg_len = nullptr, *seg_next = nullptr, *seg_tail = nullptr, *seg_chain = nullptr, *seg_off = nullptr;
    u32 *chunk_next = nullptr;
    u64 *pool = nullptr, *chain_stream = nullptr, *chain_tail = nullptr;
    bool seg_ok = false, use_stream = false, known_long = false;
    u8 *seg_marks = nullptr;   // nodes walked by the first long-chain pass (kc_segment_walk_kernel<1>)
    u64 nstarts = 0, nchains = 0;
    // (marks of the segment walks: where a walk enters a block only -- kc_frag_unvisited_kernel turns them into the unvisited k-mers;
    // blocks beyond what that kernel keeps in LDS, or KC_SEG_DENSE_MARKS: every node is marked)
    const u32 sparse_stride = (set->bshift <= 13 && !getenv("KC_SEG_DENSE_MARKS")) ? nstride : 0u;
    // Long chains over CONTRACTED links (kc_segment_hop_kernel): the in-block runs are contracted first, the segment walks and the emit
    // hop from run to run.  KC_SEG_PLAIN=1: the plain segment walks (one scattered line per k-mer).
    const bool hop_ok = set->mlen && sparse_stride && !getenv("KC_NOFRAG") && !getenv("KC_SEG_PLAIN");
    bool hop = false;   // the links are contracted right now
    u64 *hpool = nullptr, *hbase = nullptr, *hcur = nullptr, *hlog = nullptr, *hblog = nullptr;
    // one thread per segment start: mode 0 measure, 1 measure + mark, 2 the segments of no chain take their marks back
    auto seg_walk = [&](int mode, u8 *marks) -> int {
        const dim3 g(kc_grid(nstarts, 256)), b(256);
        const u64 *sc = mode == 2 ? (const u64 *)seg_chain : (const u64 *)nullptr;
        const u32 st = mode == 0 ? 0u : sparse_stride;
        if (hop) {
            if (mode == 0) KC_LAUNCH(kc_segment_hop_kernel<0>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, nstride, compl_);
            else if (mode == 1) KC_LAUNCH(kc_segment_hop_kernel<1>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, nstride, compl_);
            else KC_LAUNCH(kc_segment_hop_kernel<2>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, nstride, compl_);
        } else {
            if (mode == 0) KC_LAUNCH(kc_segment_walk_kernel<0>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, st);
            else if (mode == 1) KC_LAUNCH(kc_segment_walk_kernel<1>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, st);
            else KC_LAUNCH(kc_segment_walk_kernel<2>, g, b, 0, ctx->stream, (const u64 *)link, (const u64 *)starts, nstarts, seg_len, seg_next, seg_tail, marks, sc, NID, st);
        }
        return KC_OK;
    };
    // (more k-mers per chain head than the limit: some chain is longer than it, whatever the per-chain pass would find -- the 4 ms it
    // takes to give up on a genome are saved; k-mers on cycles only make this a cautious guess: the segment path serves short chains too)
    if (nheads && N / nheads > KC_CHAIN_LIMIT) known_long = true;
    for (int pass = 0;; ++pass) {
        const u64 max_chains = nheads + 1;
        KC_TRY(scope.alloc(&chain_head, max_chains));
        KC_TRY(scope.alloc(&chain_len, max_chains));
        KC_TRY(scope.alloc(&chain_key, max_chains * W));
        // per-chain measure pass with a step limit: read-like inputs (chains of ~1000 nodes) finish here
        u64 hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
        // recording pool of the measure pass: every walk needs at most steps/1024 + 1 chunks, all walks together make 2N steps
        KcRec rec{};
        if (want_rec) {
            rec.pool_cap = 2 * N / (32 * KC_CHUNK_WORDS) + 2 * nheads + 1024;
            KC_TRY(scope.alloc(&pool, rec.pool_cap * KC_CHUNK_WORDS));
            KC_TRY(scope.alloc(&chunk_next, rec.pool_cap));
            KC_TRY(scope.alloc(&chain_stream, max_chains));
            KC_TRY(scope.alloc(&chain_tail, max_chains));
            rec.pool = pool; rec.chunk_next = chunk_next; rec.pool_cursor = cnt + 5; rec.chain_stream = chain_stream; rec.chain_tail = chain_tail;
        }
        // fragment contraction (see kc_frag_contract_kernel): pays when the chains are short on average (read-like input);
        // KC_NOFRAG=1 turns it off
        bool use_frag = pass == 0 && want_rec && !known_long && nheads && set->mlen && set->bshift <= 13 && !getenv("KC_NOFRAG") &&
                        2 * N / nheads <= KC_CHAIN_LIMIT / 2;
        if (use_frag) {
            u64 *fpool = nullptr, *fbase = nullptr, *fcur = nullptr, *flog = nullptr, *fblog = nullptr;
            // per block (the kernel divides both evenly): 1/6 slot of log entries -- two ends per in-block run, ~1 run per 9 k-mers -- and
            // 1/16 slot of pool words (runs beyond 19 nodes: ~100 per block of 8192 slots, one or two words each)
            u64 pool_cap = (cap / 16 + 65536) / 64 * 64;
            if (pool_cap > ((u64)1 << 29)) pool_cap = (u64)1 << 29;   // (a shortcut names a pool run by its absolute first word in 29 bits; blocks without room keep plain links)
            const u64 log_cap = (cap / 6 + 65536) / 64 * 64;
            KC_TRY(scope.alloc(&fpool, pool_cap)); KC_TRY(scope.alloc(&fbase, set->nblocks)); KC_TRY(scope.alloc(&fcur, 2 * 64 * 16));
            KC_TRY(scope.alloc(&flog, log_cap)); KC_TRY(scope.alloc(&fblog, set->nblocks));
            KC_HIP(ctx, hipMemsetAsync(fcur, 0, 2 * 64 * 16 * 8, ctx->stream));
            const size_t lds = (size_t)2 * ((size_t)1 << set->bshift) * 4 + KC_FRAG_LONGS * 8 + KC_FRAG_ENTRIES * 2;
            KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_frag_contract_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            KC_LAUNCH((kc_frag_contract_kernel<W, true>), dim3((unsigned)set->nblocks), dim3(1024), lds, ctx->stream, sv, link, compl_, fpool, pool_cap, fcur, fbase,
                               flog, log_cap, fcur + 64 * 16, fblog);
            const KcFrag fr{fpool, fbase, 0, 0};
            // bidirectional sets: one launch per class of heads, so that most chains are walked from one end only (see the kernel);
            // KC_WALK_BOTH=1: every head walks, as before round 5
            // (the class passes trade walks for launches: four launches whose walks each take ~0.3 ms whatever their number -- ~110 dependent
            // hops -- against 0.27 N scattered lines saved.  Below ~1.3 10^8 k-mers the launches cost more than the lines: every head walks,
            // one launch (cfg[1]: walks 1.33 -> 0.4 ms of a 13 ms pass).  KC_WALK_CLASSES_MIN: the k-mer count from which the classes are used)
            u64 classes_min = (u64)1 << 27;
            if (const char *e = getenv("KC_WALK_CLASSES_MIN")) classes_min = (u64)atoll(e);
            if (compl_ && !getenv("KC_WALK_BOTH") && N >= classes_min) {
                u64 *hcls = nullptr, *hcnt = nullptr;
                KC_TRY(scope.alloc(&hcls, (u64)KC_WALK_CLASSES * nheads));
                KC_TRY(scope.alloc(&hcnt, KC_WALK_CLASSES));
                KC_HIP(ctx, hipMemsetAsync(hcnt, 0, KC_WALK_CLASSES * 8, ctx->stream));
                KC_LAUNCH(kc_heads_by_class_kernel, dim3(kc_grid(nheads, 1024)), dim3(1024), 0, ctx->stream, (const u64 *)heads, nheads, hcls, hcnt);
                u64 hn[KC_WALK_CLASSES];
                KC_HIP(ctx, hipMemcpyAsync(hn, hcnt, sizeof hn, hipMemcpyDeviceToHost, ctx->stream));
                KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                u64 t0 = 0;
                for (int cp = 0; cp < KC_WALK_CLASSES; ++cp) {
                    if (hn[cp])
                        KC_LAUNCH((kc_walk_measure_kernel<W, false, true, true>), dim3(kc_grid(hn[cp], 256)), dim3(256), 0, ctx->stream, sv,
                                           (const u64 *)link, compl_, (const u64 *)(hcls + (u64)cp * nheads), hn[cp], (u8 *)nullptr, chain_head, chain_len, chain_key, cnt,
                                           KC_CHAIN_LIMIT, rec, fr, link, cp, t0, nheads);
                    t0 += hn[cp];
                }
                KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (hcls belongs to this block)
                scope.release(hcls); scope.release(hcnt);
            } else
            KC_LAUNCH((kc_walk_measure_kernel<W, false, true, true>), dim3(kc_grid(nheads, 256)), dim3(256), 0, ctx->stream, sv,
                               (const u64 *)link, compl_, (const u64 *)heads, nheads, (u8 *)nullptr, chain_head, chain_len, chain_key, cnt,
                               KC_CHAIN_LIMIT, rec, fr);
            KC_HIP(ctx, hipGetLastError());
            u64 hf[8] = {0};
            KC_HIP(ctx, hipMemcpyAsync(hf, cnt, 64, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const bool clean = !hf[7] && !hf[6] && hf[3] == N;
            if (getenv("KC_SIMP_TRACE"))
                fprintf(stderr, "[simp] fragment walk: too long %llu, recording overflow %llu, covered %llu of %llu, chains %llu -> %s\n",
                        (unsigned long long)hf[7], (unsigned long long)hf[6], (unsigned long long)hf[3], (unsigned long long)N, (unsigned long long)hf[0],
                        clean ? "kept" : "links restored, classic flow");
            if (!clean) {
                // a chain beyond the limit, or k-mers on cycles (reached from no head): put the replaced link words back and
                // let the classic flow below run on the original links
                KC_LAUNCH(kc_frag_restore_kernel, dim3((unsigned)set->nblocks), dim3(256), 0, ctx->stream, link, (const u64 *)flog,
                                   (const u64 *)fblog, set->bshift, (u64)0, nstride);
                if (compl_ && !getenv("KC_WALK_BOTH")) KC_LAUNCH(kc_clear_walked_kernel, dim3(kc_grid(nheads, 256)), dim3(256), 0, ctx->stream, link, (const u64 *)heads, nheads);
                KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
                if (hf[7]) known_long = true;
                use_frag = false;
            }
            scope.release(fpool); scope.release(fbase); scope.release(fcur); scope.release(flog); scope.release(fblog);
        }
        if (nheads && !known_long && !use_frag) {
            if (want_rec)
                KC_LAUNCH((kc_walk_measure_kernel<W, false, true>), dim3(kc_grid(nheads, 256)), dim3(256), 0, ctx->stream, sv,
                                   (const u64 *)link, compl_, (const u64 *)heads, nheads, (u8 *)nullptr, chain_head, chain_len, chain_key, cnt,
                                   KC_CHAIN_LIMIT, rec);
            else
                KC_LAUNCH((kc_walk_measure_kernel<W, false, false>), dim3(kc_grid(nheads, 256)), dim3(256), 0, ctx->stream, sv,
                                   (const u64 *)link, compl_, (const u64 *)heads, nheads, (u8 *)nullptr, chain_head, chain_len, chain_key, cnt,
                                   KC_CHAIN_LIMIT, no_rec);
        }
        KC_HIP(ctx, hipMemcpyAsync(hc, cnt, 64, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const bool too_long = known_long || hc[7] != 0;
        known_long = too_long;   // (cutting cycles only adds chains: a later pass need not find that out again)
        // the recordings are used when this pass is the one that produces the chains: no long chain, no cycles, pool not exhausted
        use_stream = want_rec && !too_long && hc[3] == N && hc[6] == 0;
        if (!use_stream) {
            scope.release(pool); scope.release(chunk_next); scope.release(chain_stream); scope.release(chain_tail);
            pool = nullptr; chunk_next = nullptr; chain_stream = nullptr; chain_tail = nullptr;
        }
        nchains = hc[0];
        if (!too_long && hc[3] == N) break;
        // ---- some chain is long: segment-parallel path
        seg_ok = false;
        if (too_long) {
            if (hop_ok) {
                // contract the in-block runs (every run end stays a landing place: no_next), as kc_frag_contract_kernel is used for reads.
                // A second pass finds the contraction of the first: kc_frag_unvisited_kernel put only the blocks with cycles back to
                // plain links, and only those are contracted again (a cycle that the rule opened is a long stretch of some chain now:
                // left plain, without splitters, one thread would walk all of it).
                const u64 pool_cap = std::min<u64>((cap / 16 + 65536) / 64 * 64, (u64)1 << 29);
                const u64 log_cap = (cap / 6 + 65536) / 64 * 64;
                if (!hop) {
                    KC_TRY(scope.alloc(&hpool, pool_cap)); KC_TRY(scope.alloc(&hbase, set->nblocks)); KC_TRY(scope.alloc(&hcur, 2 * 64 * 16));
                    KC_TRY(scope.alloc(&hlog, log_cap)); KC_TRY(scope.alloc(&hblog, set->nblocks));
                    KC_HIP(ctx, hipMemsetAsync(hcur, 0, 2 * 64 * 16 * 8, ctx->stream));
                }
                const size_t lds = (size_t)2 * ((size_t)1 << set->bshift) * 4 + KC_FRAG_LONGS * 8 + KC_FRAG_ENTRIES * 2;
                KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_frag_contract_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                KC_LAUNCH((kc_frag_contract_kernel<W, true>), dim3((unsigned)set->nblocks), dim3(1024), lds, ctx->stream, sv, link, compl_, hpool, pool_cap, hcur, hbase,
                                   hlog, log_cap, hcur + 64 * 16, hblog, (u64 *)nullptr, (u64)0, (u64 *)nullptr, 1, hop ? 1 : 0);
                hop = true;
            }
            const u64 nsplit_hop = (NID + 63) >> 6;
            const u64 max_starts = nheads + 2 * (hop ? nsplit_hop : nsplit_max) + 1;
            u64 *starts_raw = nullptr;
            KC_TRY(scope.alloc(&starts_raw, max_starts));
            KC_TRY(scope.alloc(&starts, max_starts));
            KC_HIP(ctx, hipMemcpyAsync(starts_raw, heads, nheads * 8, hipMemcpyDeviceToDevice, ctx->stream));
            KC_HIP(ctx, hipMemcpyAsync(cnt + 6, &nheads, 8, hipMemcpyHostToDevice, ctx->stream));
            if (hop)
                KC_LAUNCH(kc_split_starts_hop_kernel, kc_grid2(nsplit_hop, 256), dim3(256), 0, ctx->stream, (const u64 *)link,
                                   NID, ncount, nstride, compl_, starts_raw, cnt + 6);
            else
                KC_LAUNCH(kc_split_starts_kernel, dim3(kc_grid(nsplit_max, 256)), dim3(256), 0, ctx->stream, (const u64 *)link,
                                   NID, ncount, nstride, compl_, starts_raw, cnt + 6);
            KC_TRY(read_u64(ctx, cnt + 6, &nstarts));
            int start_bits = 2;   // (a start is node id << 1 | orientation)
            while (start_bits < 64 && (NID >> (start_bits - 1)) != 0) ++start_bits;
            KC_TRY(kcu_sort_pairs(ctx, starts_raw, starts, nullptr, nullptr, nstarts, 0, start_bits));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            scope.release(starts_raw);
            if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] pass %d: long chains, %llu segment starts (%s links)\n", pass, (unsigned long long)nstarts, hop ? "contracted" : "plain");
            KC_TRY(scope.alloc(&seg_len, nstarts + 1)); KC_TRY(scope.alloc(&seg_next, nstarts + 1)); KC_TRY(scope.alloc(&seg_tail, nstarts + 1));
            KC_TRY(scope.alloc(&seg_chain, nstarts + 1)); KC_TRY(scope.alloc(&seg_off, nstarts + 1));
            KC_HIP(ctx, hipMemsetAsync(seg_chain, 0xff, (nstarts + 1) * 8, ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(cnt, 0, 32, ctx->stream));
            // the first long-chain pass marks what it walks: should the chains not cover the set (cycles), the unmarked k-mers are at
            // hand without a second walk over everything (pass 1 and later know that the cycles are cut: no marks)
            seg_marks = nullptr;
            if (pass == 0 && !getenv("KC_SEG_NOMARK")) {
                KC_TRY(scope.alloc(&seg_marks, NID));
                KC_HIP(ctx, hipMemsetAsync(seg_marks, 0, NID, ctx->stream));
            }
            if (nstarts) {
                KC_TRY(seg_walk(seg_marks ? 1 : 0, seg_marks));
                KC_LAUNCH((kc_chain_reduce_kernel<W>), dim3(kc_grid(nstarts, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, compl_,
                                   (const u64 *)starts, nstarts, (const u64 *)seg_len, (const u64 *)seg_next, (const u64 *)seg_tail, seg_chain,
                                   seg_off, chain_head, chain_len, chain_key, cnt);
            }
            KC_HIP(ctx, hipMemcpyAsync(hc, cnt, 32, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            seg_ok = (hc[3] == N);
            nchains = hc[0];
            if (seg_ok) break;
        }
        // ---- not every k-mer is on a chain with ends: cycles.  Mark what the heads reach, cut the rest, go again.
        if (pass == 2) return kc_fail(ctx, KC_E_NO_DEVICE, "chain discovery does not cover the set after cutting the cycles");
        u8 *visited = nullptr;
        bool marked = false;
        if (too_long && seg_marks) {
            // marked by the measuring walk already; the segments no chain took (their splitters lie on cycles) take their marks back
            visited = seg_marks;
            seg_marks = nullptr;
            marked = true;
            if (nstarts) KC_TRY(seg_walk(2, visited));
        } else {
            KC_TRY(scope.alloc(&visited, NID));
            KC_HIP(ctx, hipMemsetAsync(visited, 0, NID, ctx->stream));
        }
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
        if (too_long && marked) {
        } else if (too_long) {
            if (nstarts) {   // (a marking walk of its own, then the segments of no chain take their marks back)
                KC_TRY(seg_walk(1, visited));
                KC_TRY(seg_walk(2, visited));
            }
        } else if (nheads) {
            KC_LAUNCH((kc_walk_measure_kernel<W, true, false>), dim3(kc_grid(nheads, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link,
                               compl_, (const u64 *)heads, nheads, visited, chain_head, chain_len, chain_key, cnt, ~(u64)0, no_rec);
        }
        // (hop: the links stay contracted; kc_frag_unvisited_kernel works from the log and restores the blocks that hold cycles)
        KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
        const bool sparse_marks = too_long && sparse_stride;
        const size_t lds_unv = (size_t)14 << set->bshift;
        u64 nunvis = 0;
        u64 *ulist = nullptr, *heads2 = nullptr;
        if (sparse_marks) {
            // one pass into a list sized for the usual case (k-mers on cycles are few); the cursor counts on past its end
            u64 ucap = N / 64 + ((u64)1 << 20);
            KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_frag_unvisited_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_unv));
            for (int attempt = 0; attempt < 2; ++attempt) {
                KC_TRY(scope.alloc(&ulist, ucap));
                KC_HIP(ctx, hipMemsetAsync(cnt + 1, 0, 8, ctx->stream));
                KC_LAUNCH((kc_frag_unvisited_kernel<W, true>), dim3((unsigned)set->nblocks), dim3(1024), lds_unv, ctx->stream, sv, link, compl_,
                                   (const u8 *)visited, ulist, ucap, cnt + 1, hop ? (const u64 *)hlog : (const u64 *)nullptr, hop ? hblog : (u64 *)nullptr);
                KC_TRY(read_u64(ctx, cnt + 1, &nunvis));
                if (nunvis <= ucap) break;
                scope.release(ulist); ulist = nullptr;
                ucap = nunvis;
            }
        } else {
            KC_LAUNCH((kc_unvisited_kernel<false>), kc_grid2(NID, 256), dim3(256), 0, ctx->stream, NID, ncount, nstride, (const u8 *)visited,
                               (u64 *)nullptr, cnt + 1);
            KC_TRY(read_u64(ctx, cnt + 1, &nunvis));
            KC_TRY(scope.alloc(&ulist, nunvis + 1));
            KC_HIP(ctx, hipMemsetAsync(cnt + 1, 0, 8, ctx->stream));
            KC_LAUNCH((kc_unvisited_kernel<true>), kc_grid2(NID, 256), dim3(256), 0, ctx->stream, NID, ncount, nstride, (const u8 *)visited, ulist,
                               cnt + 1);
        }
        if (getenv("KC_SIMP_TRACE")) fprintf(stderr, "[simp] pass %d: %llu k-mers on cycles (%s marks)\n", pass, (unsigned long long)nunvis, sparse_marks ? "sparse" : "dense");
        if (nunvis == 0) return kc_fail(ctx, KC_E_NO_DEVICE, "chain discovery misses k-mers that are not on cycles");
        {
            // rows of the k-mers on cycles and the candidate edges of their ports to dead ports -> host, which walks the cycles and
            // decides where each is opened (kc_cycles_decide) -> kc_cycle_fix_kernel, new head list
            const int RW = 3 + W, CW = 4 + W;
            u64 *rows_d = nullptr, *cand_d = nullptr, *fix_d = nullptr;
            const u64 cand_cap = nunvis + 65536;
            KC_TRY(scope.alloc(&rows_d, nunvis * RW)); KC_TRY(scope.alloc(&cand_d, cand_cap * CW));
            KC_HIP(ctx, hipMemsetAsync(cnt + 6, 0, 8, ctx->stream));
            KC_LAUNCH((kc_cycle_rows_kernel<W, true>), kc_grid2(nunvis, 256), dim3(256), 0, ctx->stream, sv, (const u64 *)link, (const u64 *)ulist, nunvis, rows_d);
            KC_LAUNCH((kc_cycle_cand_kernel<W>), kc_grid2(2 * nunvis, 256), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, compl_, (const u64 *)ulist, nunvis,
                               cand_d, cand_cap, cnt + 6);
            KC_HIP(ctx, hipGetLastError());
            u64 ncand = 0;
            KC_TRY(read_u64(ctx, cnt + 6, &ncand));
            if (ncand > cand_cap) return kc_fail(ctx, KC_E_CAPACITY, "too many candidate edges between cycles and chain ends");
            std::vector<u64> rows(nunvis * RW), cand(ncand * CW), fix;
            KC_HIP(ctx, hipMemcpyAsync(rows.data(), rows_d, rows.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            if (ncand) KC_HIP(ctx, hipMemcpyAsync(cand.data(), cand_d, cand.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            const u64 ncyc = kc_cycles_decide<W>(rows, nunvis, cand, ncand, fix);
            if (ncyc == 0) return kc_fail(ctx, KC_E_NO_DEVICE, "k-mers outside every chain that lie on no cycle");
            const u64 nfix = fix.size() / 2;
            KC_TRY(scope.alloc(&fix_d, fix.size()));
            KC_TRY(scope.alloc(&heads2, nheads + nfix + 1));
            KC_HIP(ctx, hipMemcpyAsync(fix_d, fix.data(), fix.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
            KC_LAUNCH((kc_cycle_fix_kernel<W>), kc_grid2(nfix, 256), dim3(256), 0, ctx->stream, sv, link, (const u64 *)fix_d, nfix);
            KC_LAUNCH(kc_heads_refresh_kernel, kc_grid2(nheads + nfix, 256), dim3(256), 0, ctx->stream, (const u64 *)link, (const u64 *)heads, nheads, (const u64 *)fix_d, nfix,
                               compl_, heads2, cnt);
            KC_HIP(ctx, hipGetLastError());
            KC_TRY(read_u64(ctx, cnt, &nheads));   // (synchronises: fix is a host vector)
            cycles_cut += ncyc;
            scope.release(rows_d); scope.release(cand_d); scope.release(fix_d);
        }
        scope.release(heads); heads = heads2;
        scope.release(ulist); scope.release(visited);
        scope.release(chain_head); scope.release(chain_len); scope.release(chain_key);
        scope.release(starts); scope.release(seg_len); scope.release(seg_next); scope.release(seg_tail); scope.release(seg_chain); scope.release(seg_off);
        starts = seg_len = seg_next = seg_tail = seg_chain = seg_off = nullptr;
        nstarts = 0;
    }
    KC_HIP(ctx, hipGetLastError());
    scope.release(heads);
    scope.release(seg_marks);

    // order chains by head key, size them, emit
    const u64 S = nchains;
    kc_simplitigs *s = new kc_simplitigs();
    s->ctx = ctx; s->k = k; s->W = W; s->n = S; s->occ_ratio = set->occ_ratio;
    bool built = false;
    auto cleanup = kc_defer([&] { if (!built) kc_simplitigs_destroy(s); });
    u64 *perm = nullptr, *nwords = nullptr;
    int rc = kcp_sort_perm_by_keys(ctx, chain_key, S, W, k, &perm);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->lens, S + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->woff, S + 1);
    if (rc == KC_OK) rc = scope.alloc(&nwords, S + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->firstK, S * W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->lastK, S * W);
    if (rc != KC_OK) return rc;
    KC_LAUNCH(kc_chain_sizes_kernel, dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, (const u64 *)chain_len, (const u64 *)perm, S, k,
                       s->lens, nwords);
    KC_TRY(kcu_exclusive_scan(ctx, nwords, s->woff, S));   // (woff[S] = total)
    KC_TRY(read_u64(ctx, s->woff + S, &s->nwords));
    s->total_bases = N + S * (u64)(k - 1);
    rc = kc_alloc(ctx, &s->words, s->nwords + 1);
    if (rc != KC_OK) return rc;
    if (seg_ok) {
        u64 *rank = nullptr;
        KC_TRY(scope.alloc(&rank, S + 1));
        KC_HIP(ctx, hipMemsetAsync(s->words, 0, (s->nwords + 1) * 8, ctx->stream));
        KC_LAUNCH(kc_inverse_perm_kernel, dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, (const u64 *)perm, S, rank);
        if (hop)
            KC_LAUNCH((kc_segment_hop_emit_kernel<W>), dim3(kc_grid(nstarts, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, (const u64 *)hpool,
                               (const u64 *)starts, nstarts, (const u64 *)seg_chain, (const u64 *)seg_off, (const u64 *)rank, (const u64 *)s->woff,
                               s->words, s->firstK, s->lastK);
        else
            KC_LAUNCH((kc_segment_emit_kernel<W>), dim3(kc_grid(nstarts, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k,
                               (const u64 *)starts, nstarts, (const u64 *)seg_chain, (const u64 *)seg_off, (const u64 *)rank, (const u64 *)s->woff,
                               s->words, s->firstK, s->lastK);
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        scope.release(rank);
    } else if (use_stream) {
        // lanes per simplitig by the mean number of words (KC_STREAM_LANES: test knob, 1 / 2 / 8)
        int lanes = s->nwords >= 6 * S ? 8 : s->nwords >= 2 * S ? 2 : 1;   // (measured on 150-bp reads with errors -- 2.5 words a simplitig: 3.80 / 3.57 / 4.42 ms with 1 / 2 / 8)
        if (const char *e = getenv("KC_STREAM_LANES")) lanes = atoi(e);
        if (lanes >= 8)
            KC_LAUNCH((kc_stream_emit_kernel<W, 3>), dim3(kc_grid(S * 8, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, (const u64 *)chain_head,
                               (const u64 *)chain_len, (const u64 *)chain_stream, (const u64 *)chain_tail, (const u64 *)perm, S, (const u64 *)pool,
                               (const u32 *)chunk_next, (const u64 *)s->woff, s->words, s->firstK, s->lastK);
        else if (lanes >= 2)
            KC_LAUNCH((kc_stream_emit_kernel<W, 1>), dim3(kc_grid(S * 2, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, (const u64 *)chain_head,
                               (const u64 *)chain_len, (const u64 *)chain_stream, (const u64 *)chain_tail, (const u64 *)perm, S, (const u64 *)pool,
                               (const u32 *)chunk_next, (const u64 *)s->woff, s->words, s->firstK, s->lastK);
        else
            KC_LAUNCH((kc_stream_emit_kernel<W, 0>), dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, (const u64 *)chain_head,
                               (const u64 *)chain_len, (const u64 *)chain_stream, (const u64 *)chain_tail, (const u64 *)perm, S, (const u64 *)pool,
                               (const u32 *)chunk_next, (const u64 *)s->woff, s->words, s->firstK, s->lastK);
    } else {
        KC_LAUNCH((kc_walk_emit_kernel<W>), dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k,
                           (const u64 *)chain_head, (const u64 *)perm, S, (const u64 *)s->woff, s->words, s->firstK, s->lastK);
    }
    scope.release(starts); scope.release(seg_len); scope.release(seg_next); scope.release(seg_tail); scope.release(seg_chain);
    scope.release(seg_off);
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    scope.release(nwords); kc_release(ctx, perm);
    scope.release(chain_head); scope.release(chain_len); scope.release(chain_key);
    scope.release(link); scope.release(cnt);
    built = true;
    if (cycles_out) *cycles_out = cycles_cut;
    *out = s;
    return KC_OK;
}

int kcp_build_simplitigs(kc_set *set, kc_simplitigs **out, u64 *rounds, u64 *cycles) {
    switch (set->W) {
        case 1: return build_simplitigs_w<1>(set, out, rounds, cycles);
        case 2: return build_simplitigs_w<2>(set, out, rounds, cycles);
        default: return build_simplitigs_w<4>(set, out, rounds, cycles);
    }
}

extern "C" int kc_simplitigs_build(kc_set *set, kc_simplitigs **out) {
    if (!set || !out) return KC_E_ARG;
    *out = nullptr;
    KC_HIP(set->ctx, hipSetDevice(set->ctx->device));
    return kcp_build_simplitigs(set, out, nullptr, nullptr);
}

extern "C" void kc_simplitigs_destroy(kc_simplitigs *s) {
    if (!s) return;
    kc_ctx *ctx = s->ctx;
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, s->lens); kc_release(ctx, s->woff); kc_release(ctx, s->words);
    kc_release(ctx, s->firstK); kc_release(ctx, s->lastK);
    delete s;
}

extern "C" int kc_simplitigs_count(kc_simplitigs *s, uint64_t *n, uint64_t *total_bases) {
    if (!s) return KC_E_ARG;
    if (n) *n = s->n;
    if (total_bases) *total_bases = s->total_bases;
    return KC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// -S input: ASCII records -> packed simplitigs (simplitig_from_string, simplitigs.h:78-87).
__device__ __forceinline__ u64 kc_upper_bound(const u64 *a, u64 n, u64 v) {  // first index with a[idx] > v
    u64 lo = 0, hi = n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(256) kc_pack_records_kernel(const u8 *__restrict__ seq, const u64 *__restrict__ rec_off,
                                                              const u64 *__restrict__ lens, const u64 *__restrict__ woff, u64 n,
                                                              u64 nwords, u64 *words, u64 *bad) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwords) return;
    u64 i = kc_upper_bound(woff, n + 1, t) - 1;
    u64 b0 = (t - woff[i]) * 32, L = lens[i];
    u64 acc = 0;
    bool isbad = false;
    for (int j = 0; j < 32; ++j) {
        u64 code = 0;
        if (b0 + j < L) {
            u32 ch = seq[rec_off[i] + b0 + j];
            u32 up = ch & 0xDFu;
            bool ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
            if (!ok) isbad = true;
            code = (up >> 1) & 3u;
            code ^= code >> 1;
        }
        acc = (acc << 2) | code;
    }
    words[t] = acc;
    if (isbad) atomicExch((ull *)bad, 1ULL);
}

template <int W>
__device__ __forceinline__ Key<W> kc_extract_kmer(const u64 *words, u64 base, int k) {  // bases [base, base+k) of a packed stream
    Key<W> r = key_zero<W>();
    for (int i = 0; i < k; ++i) {
        u64 p = base + i;
        u64 b = (words[p >> 5] >> (62 - 2 * (p & 31))) & 3;
        r = key_append(r, b, k);
    }
    return r;
}

template <int W>
__global__ void __launch_bounds__(256) kc_end_kmers_kernel(const u64 *__restrict__ words, const u64 *__restrict__ woff,
                                                           const u64 *__restrict__ lens, u64 n, int k, u64 *firstK, u64 *lastK) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 b = woff[t] * 32;
    Key<W> f = kc_extract_kmer<W>(words, b, k), l = kc_extract_kmer<W>(words, b + lens[t] - k, k);
    for (int i = 0; i < W; ++i) { firstK[t * W + i] = f.w[i]; lastK[t * W + i] = l.w[i]; }
}

// positions of the bytes equal to '\n' in a device sequence, ascending (two passes over 16 KB tiles: count, scan, write)
template <bool FILL>
__global__ void __launch_bounds__(256) kc_sep_positions_kernel(const u8 *__restrict__ seq, u64 len, const u64 *__restrict__ tile_off, u64 *out) {
    const u64 base = ((u64)blockIdx.y * gridDim.x + blockIdx.x) * 16384 + (u64)threadIdx.x * 64;
    u64 m = 0;
    if (base < len) {
        const u64 n = len - base < 64 ? len - base : 64;
        for (u64 i = 0; i < n; ++i) m |= (u64)(seq[base + i] == (u8)'\n') << i;
    }
    u64 tot;
    const u64 ex = kc_block_excl_scan((u64)__popcll(m), &tot);
    const u64 tile = (u64)blockIdx.y * gridDim.x + blockIdx.x;
    if (!FILL) { if (threadIdx.x == 0) out[tile] = tot; return; }
    u64 pos = tile_off[tile] + ex;
    while (m) { out[pos++] = base + (u64)__builtin_ctzll(m); m &= m - 1; }
}
// Record starts of a device sequence whose records are separated by '\n' (what kc_read_fasta_device leaves behind: one separator per
// header line, the first of them in front): offs[i] = first character of record i, offs[n] = one past the last record's end + 1, so that
// record i has offs[i + 1] - offs[i] - 1 characters.  An empty stretch in front of the first separator or behind the last is no record.
int kcp_record_starts_from_separators(kc_ctx *ctx, const u8 *d_seq, u64 len, std::vector<u64> &offs) {
    KcScope scope(ctx);
    offs.clear();
    if (len == 0) return KC_OK;
    const u64 ntiles = (len + 16383) / 16384;
    u64 *tcount = nullptr, *pos = nullptr;
    KC_TRY(scope.alloc(&tcount, ntiles + 2));
    const dim3 g = kc_grid2(ntiles * 256, 256);   // (one workgroup per tile)
    KC_LAUNCH((kc_sep_positions_kernel<false>), g, dim3(256), 0, ctx->stream, d_seq, len, (const u64 *)nullptr, tcount);
    KC_TRY(kcu_exclusive_scan(ctx, tcount, tcount, ntiles));
    u64 nsep = 0;
    KC_TRY(read_u64(ctx, tcount + ntiles, &nsep));
    std::vector<u64> seps(nsep);
    if (nsep) {
        KC_TRY(scope.alloc(&pos, nsep + 1));
        KC_LAUNCH((kc_sep_positions_kernel<true>), g, dim3(256), 0, ctx->stream, d_seq, len, (const u64 *)tcount, pos);
        KC_HIP(ctx, hipGetLastError());
        KC_HIP(ctx, hipMemcpyAsync(seps.data(), pos, nsep * 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // records = the stretches between consecutive separators (and the two ends)
    offs.reserve(nsep + 2);
    u64 start = 0;   // first character of the record at hand
    for (u64 i = 0; i <= nsep; ++i) {
        const u64 end = i < nsep ? seps[i] : len;   // one past its last character
        const bool edge = (i == 0 || i == nsep) && end == start;   // nothing in front of the first / behind the last separator
        if (!edge) offs.push_back(start);
        start = end + 1;
    }
    if (!offs.empty()) offs.push_back(len + 1);
    // (interior records keep their true extents: offs[i + 1] - offs[i] - 1 is right because every record but the last is followed by
    // exactly one separator, and the last ends at len = (len + 1) - 1)
    return KC_OK;
}

int kcp_simplitigs_from_device_ascii(kc_ctx *ctx, int k, const u8 *d_seq, u64 len, const u64 *h_offsets, u64 n,
                                     int offsets_are_host_positions, kc_simplitigs **out, int gap) {
    (void)len; (void)offsets_are_host_positions;
    if (n == 0) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
    const int W = kc_words_for_k(k);
    std::vector<u64> lens(n), woff(n + 1);
    u64 total = 0, w = 0;
    for (u64 i = 0; i < n; ++i) {
        lens[i] = h_offsets[i + 1] - h_offsets[i] - (u64)gap;   // (gap 1: the records are followed by a separator each, kcp_record_starts_from_separators)
        if (lens[i] < (u64)k) return kc_fail(ctx, KC_E_ARG, "simplitig record shorter than k");
        woff[i] = w;
        w += lens[i] / 32 + 1;
        total += lens[i];
    }
    woff[n] = w;
    kc_simplitigs *s = new kc_simplitigs();
    s->ctx = ctx; s->k = k; s->W = W; s->n = n; s->total_bases = total; s->nwords = w;
    u64 *d_recoff = nullptr, *bad = nullptr;
    int rc = kc_alloc(ctx, &s->lens, n + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->woff, n + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->words, w + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->firstK, n * W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->lastK, n * W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &d_recoff, n + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &bad, 1);
    if (rc != KC_OK) { kc_simplitigs_destroy(s); return rc; }
    KC_HIP(ctx, hipMemcpyAsync(s->lens, lens.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipMemcpyAsync(s->woff, woff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipMemcpyAsync(d_recoff, h_offsets, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(bad, 0, 8, ctx->stream));
    KC_LAUNCH(kc_pack_records_kernel, dim3(kc_grid(w, 256)), dim3(256), 0, ctx->stream, d_seq, (const u64 *)d_recoff,
                       (const u64 *)s->lens, (const u64 *)s->woff, n, w, s->words, bad);
    unsigned g = kc_grid(n, 256);
    switch (W) {
        case 1: KC_LAUNCH((kc_end_kmers_kernel<1>), dim3(g), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff, (const u64 *)s->lens, n, k, s->firstK, s->lastK); break;
        case 2: KC_LAUNCH((kc_end_kmers_kernel<2>), dim3(g), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff, (const u64 *)s->lens, n, k, s->firstK, s->lastK); break;
        default: KC_LAUNCH((kc_end_kmers_kernel<4>), dim3(g), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff, (const u64 *)s->lens, n, k, s->firstK, s->lastK); break;
    }
    KC_HIP(ctx, hipGetLastError());
    u64 hbad = 0;
    rc = read_u64(ctx, bad, &hbad);
    kc_release(ctx, d_recoff); kc_release(ctx, bad);
    if (rc != KC_OK) { kc_simplitigs_destroy(s); return rc; }
    if (hbad) { kc_simplitigs_destroy(s); return kc_fail(ctx, KC_E_INVALID_CHAR, "simplitig input holds a non-ACGT character"); }
    *out = s;
    return KC_OK;
}

extern "C" int kc_simplitigs_from_strings(kc_ctx *ctx, int k, const char *seqs, const uint64_t *offsets, uint64_t n, kc_simplitigs **out) {
    if (!ctx || !out || (!seqs && n) || (!offsets && n)) return KC_E_ARG;
    *out = nullptr;
    if (k < 1 || k > 127) return kc_fail(ctx, KC_E_ARG, "k must be in 1..127");
    if (n == 0) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    u64 len = offsets[n];
    u8 *d = nullptr;
    KC_TRY(kc_alloc(ctx, &d, len + 64));
    KC_HIP(ctx, hipMemcpyAsync(d, seqs, len, hipMemcpyHostToDevice, ctx->stream));
    int rc = kcp_simplitigs_from_device_ascii(ctx, k, d, len, offsets, n, 1, out);
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d);
    return rc;
}

// ASCII export
__global__ void __launch_bounds__(256) kc_unpack_kernel(const u64 *__restrict__ words, const u64 *__restrict__ woff,
                                                        const u64 *__restrict__ lens, const u64 *__restrict__ out_off, u64 n, char *out) {
    // one wave per simplitig
    u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if (wave >= n) return;
    u64 L = lens[wave], wb = woff[wave], ob = out_off[wave];
    for (u64 j = lane; j < L; j += 64) {
        u64 b = (words[wb + (j >> 5)] >> (62 - 2 * (j & 31))) & 3;
        out[ob + j] = "ACGT"[b];
    }
}

extern "C" int kc_simplitigs_export(kc_simplitigs *s, char *seqs, uint64_t *offsets) {
    if (!s || !seqs || !offsets) return KC_E_ARG;
    kc_ctx *ctx = s->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<u64> lens(s->n);
    KC_HIP(ctx, hipMemcpyAsync(lens.data(), s->lens, s->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    u64 o = 0;
    for (u64 i = 0; i < s->n; ++i) { offsets[i] = o; o += lens[i]; }
    offsets[s->n] = o;
    u64 *d_off = nullptr; char *d_out = nullptr;
    KC_TRY(kc_alloc(ctx, &d_off, s->n + 1));
    KC_TRY(kc_alloc(ctx, &d_out, o + 1));
    KC_HIP(ctx, hipMemcpyAsync(d_off, offsets, (s->n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_LAUNCH(kc_unpack_kernel, dim3(kc_grid(s->n * 64, 256)), dim3(256), 0, ctx->stream, (const u64 *)s->words, (const u64 *)s->woff,
                       (const u64 *)s->lens, (const u64 *)d_off, s->n, d_out);
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipMemcpyAsync(seqs, d_out, o, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, d_off); kc_release(ctx, d_out);
    return KC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// fill_kmers (simplitigs.h:41-51): insert every k-mer of the packed simplitigs into a set. The packed words are a
// valid 2-bit stream for the insert kernel once the padding bases are marked as breaks.
__global__ void __launch_bounds__(256) kc_padding_breaks_kernel(const u64 *__restrict__ woff, const u64 *__restrict__ lens, u64 n,
                                                                u64 nwords, u64 *brk, u64 nbrk) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbrk) return;
    u64 m = 0;
    for (int half = 0; half < 2; ++half) {
        u64 cw = 2 * t + half;
        u64 bits = 0xffffffffull;  // all 32 positions break
        if (cw < nwords) {
            u64 i = kc_upper_bound(woff, n + 1, cw) - 1;
            u64 start = (cw - woff[i]) * 32, L = lens[i];
            u64 valid = L > start ? (L - start > 32 ? 32 : L - start) : 0;
            bits = valid >= 32 ? 0 : (0xffffffffull & ~((1ull << valid) - 1));
        }
        m |= bits << (32 * half);
    }
    brk[t] = m;
}

int kcp_insert_simplitig_kmers(kc_set *set, kc_simplitigs *s) {
    kc_ctx *ctx = set->ctx;
    u64 nbrk = (s->nwords + 1) / 2;
    u64 *brk = nullptr, *low = nullptr;
    KC_TRY(kcs_reserve(set, s->total_bases));
    KC_TRY(kc_alloc(ctx, &brk, nbrk + 1));
    KC_TRY(kc_alloc(ctx, &low, nbrk + 1));
    KC_HIP(ctx, hipMemsetAsync(low, 0, (nbrk + 1) * 8, ctx->stream));
    KC_LAUNCH(kc_padding_breaks_kernel, dim3(kc_grid(nbrk, 256)), dim3(256), 0, ctx->stream, (const u64 *)s->woff,
                       (const u64 *)s->lens, s->n, s->nwords, brk, nbrk);
    int rc = kcs_insert_packed(set, s->words, brk, low, s->nwords * 32, 0);
    if (rc == KC_OK) rc = kcs_refresh_size(set);
    kc_release(ctx, brk); kc_release(ctx, low);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
// Sparse branch vertex list: simplitigs_to_kmer_vec + PartialPreSort (simplitigs.h:53-65, global_sparse.h:14-35).
template <int W>
__global__ void __launch_bounds__(256) kc_expand_kmers_kernel(const u64 *__restrict__ words, const u64 *__restrict__ woff,
                                                              const u64 *__restrict__ lens, const u64 *__restrict__ koff, u64 n, int k,
                                                              u64 nk, u64 *kmers, u64 *sortkey) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk) return;
    u64 i = kc_upper_bound(koff, n + 1, t) - 1;
    Key<W> x = kc_extract_kmer<W>(words, woff[i] * 32 + (t - koff[i]), k);
    for (int j = 0; j < W; ++j) kmers[t * W + j] = x.w[j];
    int bits = 2 * k < 8 ? 2 * k : 8;
    Key<W> top = key_shr(x, 2 * k - bits);
    sortkey[t] = top.w[0] & ((1ull << bits) - 1);
    (void)lens;
}
__global__ void __launch_bounds__(256) kc_kmer_counts_kernel(const u64 *__restrict__ lens, u64 n, int k, u64 *cnt) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) cnt[t] = lens[t] - (u64)k + 1;
}
template <int W>
__global__ void __launch_bounds__(256) kc_sparse_fill_kernel(const u64 *__restrict__ kmers, const u64 *__restrict__ perm, u64 nk, int k,
                                                             int wpk, u64 *lens, u64 *woff, u64 *words, u64 *firstK, u64 *lastK) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nk) return;
    if (t == nk) { woff[nk] = nk * (u64)wpk; return; }
    u64 src = perm[t];
    Key<W> x;
    for (int j = 0; j < W; ++j) { x.w[j] = kmers[src * W + j]; firstK[t * W + j] = x.w[j]; lastK[t * W + j] = x.w[j]; }
    lens[t] = (u64)k;
    woff[t] = t * (u64)wpk;
    u64 acc = 0; int nacc = 0; u64 wp = t * (u64)wpk;
    for (int i = 0; i < k; ++i) {
        acc = (acc << 2) | key_at(x, k, i);
        if (++nacc == 32) { words[wp++] = acc; acc = 0; nacc = 0; }
    }
    words[wp] = nacc ? (acc << (64 - 2 * nacc)) : 0;
}

template <int W> static int to_sparse_w(kc_simplitigs *s, kc_simplitigs **out) {
    kc_ctx *ctx = s->ctx;
    const int k = s->k;
    u64 n = s->n;
    u64 *cnt = nullptr, *koff = nullptr;
    KC_TRY(kc_alloc(ctx, &cnt, n + 1));
    KC_TRY(kc_alloc(ctx, &koff, n + 1));
    KC_LAUNCH(kc_kmer_counts_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, (const u64 *)s->lens, n, k, cnt);
    KC_TRY(kcu_exclusive_scan(ctx, cnt, koff, n));   // (koff[n] = total)
    u64 nk = 0;
    KC_TRY(read_u64(ctx, koff + n, &nk));
    if (nk >= ((u64)1 << 31)) return kc_fail(ctx, KC_E_ARG, "more than 2^31 k-mers in the sparse branch (main.cpp:175-181) are not supported");
    u64 *kmers = nullptr, *sk = nullptr, *sk2 = nullptr, *perm = nullptr, *perm2 = nullptr;
    KC_TRY(kc_alloc(ctx, &kmers, nk * W));
    KC_TRY(kc_alloc(ctx, &sk, nk)); KC_TRY(kc_alloc(ctx, &sk2, nk));
    KC_TRY(kc_alloc(ctx, &perm, nk)); KC_TRY(kc_alloc(ctx, &perm2, nk));
    KC_LAUNCH((kc_expand_kmers_kernel<W>), dim3(kc_grid(nk, 256)), dim3(256), 0, ctx->stream, (const u64 *)s->words,
                       (const u64 *)s->woff, (const u64 *)s->lens, (const u64 *)koff, n, k, nk, kmers, sk);
    KC_LAUNCH(kc_iota_kernel, dim3(kc_grid(nk, 256)), dim3(256), 0, ctx->stream, perm, nk);
    KC_TRY(kcu_sort_pairs(ctx, sk, sk2, perm, perm2, nk, 0, 8));
    kc_simplitigs *r = new kc_simplitigs();
    int wpk = k / 32 + 1;
    r->ctx = ctx; r->k = k; r->W = W; r->n = nk; r->total_bases = nk * (u64)k; r->nwords = nk * (u64)wpk;
    int rc = kc_alloc(ctx, &r->lens, nk + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &r->woff, nk + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &r->words, r->nwords + 1);
    if (rc == KC_OK) rc = kc_alloc(ctx, &r->firstK, nk * W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &r->lastK, nk * W);
    if (rc != KC_OK) { kc_simplitigs_destroy(r); return rc; }
    KC_LAUNCH((kc_sparse_fill_kernel<W>), dim3(kc_grid(nk + 1, 256)), dim3(256), 0, ctx->stream, (const u64 *)kmers,
                       (const u64 *)perm2, nk, k, wpk, r->lens, r->woff, r->words, r->firstK, r->lastK);
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, cnt); kc_release(ctx, koff); kc_release(ctx, kmers);
    kc_release(ctx, sk); kc_release(ctx, sk2); kc_release(ctx, perm); kc_release(ctx, perm2);
    *out = r;
    return KC_OK;
}

int kcp_to_sparse(kc_simplitigs *s, kc_simplitigs **out) {
    switch (s->W) {
        case 1: return to_sparse_w<1>(s, out);
        case 2: return to_sparse_w<2>(s, out);
        default: return to_sparse_w<4>(s, out);
    }
}
extern "C" int kc_simplitigs_to_sparse(kc_simplitigs *s, kc_simplitigs **out) {
    if (!s || !out) return KC_E_ARG;
    *out = nullptr;
    KC_HIP(s->ctx, hipSetDevice(s->ctx->device));
    return kcp_to_sparse(s, out);
}
