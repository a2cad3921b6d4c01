// Hash-sharded multi-GPU flow of the BIGREEDY path: one rank per GPU, the k-mer set partitioned by table-block
// range (block = hash of the canonical minimizer, so ~90 % of a k-mer's de Bruijn neighbours live on its own rank).
// The reference has no counterpart (single thread, main.cpp:122-212); the contract is: ONE masked superstring,
// byte-identical to the single-GPU kc_compute on the union of the ranks' records.  That holds by construction: the
// canonical simplitig rule (kc_simplitigs.hip) is layout- and schedule-free, and phases 2-3 are functions of
// the ordered simplitig list.
//
//   phase 0  kcs_build_sharded (kc_set.hip): k-mers bucketed by block, all-to-all of the buckets, LDS build on the owner.
//   phase 1a kc_shard_choose_kernel: the 8 neighbour lookups per k-mer; candidates whose home block belongs to another
//            rank become queries (key + home), all-to-all, kc_shard_answer_kernel on the owner, answers back;
//            kc_shard_resolve_kernel settles the ports that waited.  Ports with >= 2 existing candidates ("branch
//            ports") contribute their edges to a small replicated subgraph that every rank settles identically by the
//            sequential form of the rule (edges by descending priority) -- branch ports are rare on read-like input.
//   phase 1b chains: every rank contracts the chain pieces that stay inside it ("fragments", kc_shard_walk_kernel, bases
//            recorded 2 bits per step); the fragments are ranked across ranks -- by SAMPLED WAVES where chains are long in
//            fragments (waves that stop at splitters, pointer jumping over the splitters only, one pull per fragment: ~40
//            single-collective rounds whatever the chain length, kc_shard_sr_*), by plain waves where they are short (one
//            round per fragment of the longest chain), by pointer jumping on unidirectional sets; chain heads gather their
//            tails and decide orientation; the chain records (head/tail k-mers, length) are all-gathered and sorted by head
//            key on every rank = the simplitig list.
//   phase 2  overlap path on the replicated list of end k-mers (V ~ 2 * 10^-3 N): the table levels with more than 2^20 live
//            vertices are divided among the ranks by key, every other level runs whole on every rank without a collective.
//   phase 3  every simplitig is emitted by the rank that owns the output range it starts in: the winning fragments send
//            their recorded bases (32 per message) there, the owner assembles the packed simplitigs and runs the ordinary
//            emit kernel into its slice of the superstring.
#include "kc_internal.h"
#include "kc_comm.h"
#include "kc_simp_dev.h"
#include "kc_scan_dev.h"
#include <algorithm>
#include <chrono>
#include <unordered_map>
#include <unordered_set>

#define KC_GID(rank, idx) (((u64)(rank) << 40) | (u64)(idx))
#define KC_GID_RANK(g) ((int)((g) >> 40))
#define KC_GID_IDX(g) ((g) & (((u64)1 << 40) - 1))
#define KC_FCW 4        /* words per recording chunk of a fragment walk (128 bases) */
#define KC_MAX_RANKS_SH 64

namespace {

// destination rank of every query (owner of the home slot)
template <int W>
__global__ void __launch_bounds__(256) kc_shard_qdest_kernel(const u64 *__restrict__ qry, u64 n, u64 rcap, u8 *dest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dest[t] = (u8)(qry[t * (W + 2) + W + 1] / rcap);
}
// the owner's side: query {x, pid of the asking port} -> that port's link word (lookups of its 4 candidates here, where they live)
template <int W>
__global__ void __launch_bounds__(256) kc_shard_answer_kernel(SetView<W> set, int k, int complements, const u64 *__restrict__ q, u64 n, u64 *ans, ShLists L) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Key<W> x;
    for (int w = 0; w < W; ++w) x.w[w] = q[t * (W + 1) + w];
    const u64 pid = q[t * (W + 1) + W] & (((u64)1 << 48) - 1);   // the asking port by dense node id (the bits above: the slot offset of its k-mer, for the asker's fill kernel)
    const int port = (int)(pid & 1);
    const Key<W> xr = key_rc(x, k);
    const Key<W> lm = key_lowmask<W>(2 * (k - 2));
    const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
    const KcPortDesc d = kc_port_desc<W, 0>(set, x, xr, k, complements, port, kc_mid_min<W, 0>(mid, midr, k, set.m));
    u64 cs[4]; u32 qb;
    if (!kc_port_search<W, false, false>(set, k, complements, x, xr, KC_NONE64, port, d, nullptr, nullptr, cs, qb))
        kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (cs[c] != KC_NONE64) cs[c] = set_nid_of_slot(set, cs[c]);   // (the link words carry dense node ids)
    ans[t] = kc_shard_settle_port<W>(k, complements, pid, x, xr, cs, qb, L);
}
// the asker's side: the answered word goes into its port; a dead port gets the slot of its k-mer, every word the slot offset (LK_OWN)
__global__ void __launch_bounds__(256) kc_shard_fill_kernel(const u64 *__restrict__ qry, const u64 *__restrict__ ans, u64 n, int QW, u64 *link, u32 nstride, int bshift) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 pw = qry[t * (u64)QW + QW - 2], pid = pw & (((u64)1 << 48) - 1), so = pw >> 48;
    u64 w = ans[t];
    if (LK_STATE(w) == LK_DEAD) w = LK_MAKE(LK_DEAD, 0, 0, (((pid >> 1) / nstride) << bshift) | so);
    link[pid] = w | LK_OWN_BITS(so);
}
// decided link words {pid, word} (branch subgraph, cycles): the ports this rank holds; the slot offset of the port's k-mer stays, a
// port that dies gets its slot
template <int W>
__global__ void __launch_bounds__(256) kc_shard_apply_fix_kernel(SetView<W> set, u64 *link, const u64 *__restrict__ fix, u64 n, u64 rlo2, u64 rcap2) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 pid = fix[2 * t];
    if (pid - rlo2 >= rcap2) return;
    const u64 old = link[pid];
    // (a word that the contraction replaced and kc_frag_restore_kernel put back has lost the offset: look the slot up)
    const u64 off = (old & LK_OWN_VALID) ? LK_OWN(old) : (set_slot_of_nid(set, pid >> 1) & (((u64)1 << set.bshift) - 1));
    u64 w = fix[2 * t + 1] & ~LK_OWN_MASK;
    if (LK_STATE(w) == LK_DEAD) w |= ((((pid >> 1) / set.nstride) << set.bshift) | off) << 6;
    link[pid] = w | LK_OWN_BITS(off);
}

// ---------------------------------------------------------------------------------------------------- phase 1b: fragments
// entry = oriented node (slot, o) whose entry port o is dead (a chain head) or matched to a slot of another rank
// (two passes: count, then fill at the scanned offsets.  The count pass keeps every thread's 16-bit entry mask, so the fill pass
// does not sweep the 16 bytes per slot of link words a second time.)
// (rlo, rcap: the NODE IDS of this rank -- block * nstride + rank in the block; the first ncount[block] ids of a block are in use)
template <bool FILL>
__global__ void __launch_bounds__(256) kc_shard_entries_kernel(const u64 *__restrict__ link, const u32 *__restrict__ ncount, u32 nstride, u64 rlo, u64 rcap, int complements,
                                                               const u64 *__restrict__ tile_off, u64 *out, unsigned short *masks) {
    const u64 base = rlo + (u64)blockIdx.x * 2048 + (u64)threadIdx.x * 8;
    u64 c = 0;
    u32 m = 0;
    if (FILL) { m = masks[(u64)blockIdx.x * 256 + threadIdx.x]; c = (u64)__popc(m); }
    else {
        for (int i = 0; i < 8; ++i) {
            const u64 slot = base + i;
            if (slot >= rlo + rcap || (u32)(slot % nstride) >= ncount[slot / nstride]) continue;
            for (int o = 0; o < (complements ? 2 : 1); ++o) {
                const u64 w = link[(slot << 1) | (u64)o];
                const bool entry = LK_STATE(w) == LK_DEAD || (LK_IS_MATCHED(w) && LK_SLOT(w) - rlo >= rcap);
                if (entry) { m |= 1u << (2 * i + o); ++c; }
            }
        }
        masks[(u64)blockIdx.x * 256 + threadIdx.x] = (unsigned short)m;
    }
    u64 tot;
    const u64 ex = kc_block_excl_scan(c, &tot);
    if (!FILL) { if (threadIdx.x == 0) out[blockIdx.x] = tot; return; }
    u64 pos = tile_off[blockIdx.x] + ex;
    for (int j = 0; j < 16; ++j)
        if ((m >> j) & 1) out[pos++] = ((base + (u64)(j >> 1)) << 1) | (u64)(j & 1);
}

struct FragRec {
    u64 *pool;        // chunks of KC_FCW words
    u32 *chunk_next;
    u64 pool_cap;
    u64 *pool_cursor;
};
// one thread per entry: walk the links while they stay on this rank; record the appended bases.  FRAG: the in-block runs have
// been contracted (kc_frag_contract_kernel over this rank's blocks): a run is crossed through the shortcut in its end node's
// inward link word -- one or two scattered lines per ~9 nodes instead of one per node.
template <bool MARK, bool FRAG>
__global__ void __launch_bounds__(256) kc_shard_walk_kernel(const u64 *__restrict__ link, u64 rlo, u64 rcap, const u64 *__restrict__ entries, u64 n,
                                                            u64 *f_exit, u64 *f_partner, u64 *f_len, u32 *f_stream, u32 *fidx, FragRec rec, u8 *visited,
                                                            u64 *flags, KcFrag frag, int bshift) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 h = entries[t];
    u64 cur = h >> 1;
    int co = (int)(h & 1);
    u64 len = 1, acc = 0, partner = KC_NONE64;
    int nacc = 0, wi = 0;
    u32 chunk = KC_NO_CHUNK, first_chunk = KC_NO_CHUNK;
    bool ok = true;
    u64 batch_next = 0;
    int batch_left = 0;
    auto flush_word = [&](u64 word) {
        if (chunk == KC_NO_CHUNK) {
            // the first chunk of walk t is chunk t: no reservation.  (Walks that hop over contracted runs reach their flushes in
            // different loop trips, so a shared cursor here means one same-address atomic per walk, ~10 ns each: measured 79 ms
            // against 43 ms for the plain walk of 10^9 k-mers before this.)
            chunk = first_chunk = (u32)t;
            wi = 0;
        } else if (wi == KC_FCW) {
            // later chunks are reserved eight at a time (a walk longer than one chunk is usually much longer: whole reads when one
            // rank holds everything)
            if (batch_left == 0) {
                batch_next = n + atomicAdd((ull *)rec.pool_cursor, 8ULL);
                batch_left = 8;
            }
            const u64 nc = batch_next++;
            --batch_left;
            if (nc >= rec.pool_cap) { ok = false; return; }
            rec.chunk_next[chunk] = (u32)nc;
            chunk = (u32)nc;
            wi = 0;
        }
        rec.pool[(u64)chunk * KC_FCW + (u64)wi++] = word;
    };
    // append nb (1..32) bases given right-aligned in `bits` (first base in the highest pair) to the recording
    auto push_group = [&](u64 bits, int nb) {
        const int room = 32 - nacc;
        if (nb < room) { acc = (acc << (2 * nb)) | bits; nacc += nb; return; }
        const int rest = nb - room;
        const u64 hi = rest ? (bits >> (2 * rest)) : bits;
        const u64 word = room == 32 ? hi : ((acc << (2 * room)) | hi);
        if (ok) flush_word(word);
        acc = rest ? (bits & (((u64)1 << (2 * rest)) - 1)) : 0;
        nacc = rest;
    };
    int xp;
    while (true) {
        if (MARK) visited[cur - rlo] = 1;
        xp = co == 0 ? 1 : 0;
        u64 w = link[(cur << 1) | (u64)xp];
        if (FRAG && KC_SC_IS(w)) {
            if (w & KC_SC_NEXT) {   // the run and the step out of it (to a slot of this rank) in one word
                const u64 nb1 = ((w >> 45) & 7) + 1;
                push_group(((((w >> 48) & 0x3fffull) >> (14 - 2 * (nb1 - 1))) << 2) | ((w >> 43) & 3), (int)nb1);
                len += nb1;
                cur = (w >> 6) & (((u64)1 << 36) - 1);
                co = (int)((w >> 42) & 1);
                continue;
            }
            u64 nb;
            if (w & KC_SC_EXT) {
                nb = (w >> 21) & 0x3fff;
                const u64 *src = frag.pool + (w >> 35);   // (dense node ids: the pool run by its absolute first word)
                for (u64 i = 0; i < nb; i += 32) {
                    const int mm = nb - i >= 32 ? 32 : (int)(nb - i);
                    push_group(src[i >> 5] >> (64 - 2 * mm), mm);
                }
            } else {
                nb = (w >> 21) & 31;
                if (nb) push_group(w >> (64 - 2 * nb), (int)nb);
            }
            len += nb;
            cur += (u64)(long long)((int)(((u32)(w >> 6) & 0x3fffu) ^ 0x2000u) - 0x2000);   // the far node: a signed 14-bit distance in ids
            co = (int)((w >> 20) & 1);
            xp = co == 0 ? 1 : 0;
            w = link[(cur << 1) | (u64)xp];   // the far end's way out of the run: a plain word
        }
        if (!LK_IS_MATCHED(w)) break;
        if (LK_SLOT(w) - rlo >= rcap) { partner = LK_PID(w); break; }
        if (!MARK) push_group(LK_BASE(w), 1);
        cur = LK_SLOT(w);
        co = LK_PORT(w) == 0 ? 0 : 1;
        ++len;
    }
    if (MARK) return;
    if (nacc && ok) flush_word(acc << (64 - 2 * nacc));
    if (!ok) flags[0] = 1;
    f_exit[t] = (cur << 1) | (u64)xp;
    f_partner[t] = partner;
    f_len[t] = len;
    f_stream[t] = first_chunk;
    fidx[h - 2 * rlo] = (u32)t;
}

// ---- pointer jumping over fragments.  nxt = global id of the next fragment (KC_NONE64: this one ends the chain);
// dist = nodes from this fragment's first node up to (not including) nxt's first node, or to the chain end; last = the
// chain's final fragment once known.
__global__ void __launch_bounds__(256) kc_shard_jump_init_kernel(const u64 *__restrict__ f_partner, const u64 *__restrict__ f_len, u64 n, int me, u64 *nxt,
                                                                 u64 *dist, u64 *last) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    nxt[t] = f_partner[t];   // still a port id: resolved to a fragment id by the first request round
    dist[t] = f_len[t];
    last[t] = f_partner[t] == KC_NONE64 ? KC_GID(me, t) : KC_NONE64;
}
// active = fragments that still have a successor; req = what to ask its owner (round 0: the entry port; later: the fragment index)
__global__ void __launch_bounds__(256) kc_shard_active_flags_kernel(const u64 *__restrict__ nxt, u64 n, u8 *flags) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flags[t] = nxt[t] != KC_NONE64;
}
__global__ void __launch_bounds__(256) kc_shard_iota_kernel(u64 *a, u64 n) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[t] = t;
}
__global__ void __launch_bounds__(256) kc_shard_req_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ nxt, int round0, u64 rcap2, u64 *req, u8 *dest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 v = nxt[act[t]];
    req[t] = v;
    dest[t] = round0 ? (u8)(v / rcap2) : (u8)KC_GID_RANK(v);
}
__global__ void __launch_bounds__(256) kc_shard_ans_port_kernel(const u64 *__restrict__ req, u64 n, const u32 *__restrict__ fidx, u64 rlo2, int me, u64 *ans) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) ans[t] = KC_GID(me, fidx[req[t] - rlo2]);
}
__global__ void __launch_bounds__(256) kc_shard_ans_jump_kernel(const u64 *__restrict__ req, u64 n, const u64 *__restrict__ nxt, const u64 *__restrict__ dist,
                                                                const u64 *__restrict__ last, u64 *ans) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    // two words: {successor, distance}; a fragment that has reached its chain's end sends the final fragment in place of the
    // successor, marked by bit 63 (ids use 46 bits)
    const u64 g = KC_GID_IDX(req[t]);
    const u64 nx = nxt[g];
    ans[2 * t] = nx == KC_NONE64 ? (last[g] | ((u64)1 << 63)) : nx;
    ans[2 * t + 1] = dist[g];
}
__global__ void __launch_bounds__(256) kc_shard_apply_port_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ ans, u64 *nxt) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) nxt[act[t]] = ans[t];
}
__global__ void __launch_bounds__(256) kc_shard_apply_jump_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ ans, u64 *nxt, u64 *dist, u64 *last) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = act[t];
    const u64 a = ans[2 * t];
    const bool ended = (a >> 63) != 0;
    nxt[e] = ended ? KC_NONE64 : a;
    dist[e] += ans[2 * t + 1];
    if (ended) last[e] = a & ~((u64)1 << 63);
}

// ---- fragment ranking by waves (bidirectional sets, chains of moderate length in fragments: read sets).  Pointer jumping
// keeps EVERY fragment exchanging 24 bytes for log2(chain length) rounds -- ~160 bytes per fragment, the largest item of the
// whole exchange.  A wave costs each fragment one 24-byte message: every chain head starts one, a fragment that receives
// {its entry port, payload, nodes before it} notes both and passes {successor port, payload, nodes + own length} on.  Both
// directions of every chain run at once, and what a fragment needs of its own direction -- the nodes from here to the chain's
// end and the chain's final fragment -- is what its TWIN (the same run walked the other way, entered at this one's exit port)
// learnt from the wave of the opposite direction: the nodes before the twin, and the payload, which the head of that wave set
// to its own twin = the final fragment of this direction.  Rounds = fragments of the longest chain; inputs with long chains
// (genomes) stay with pointer jumping.
__global__ void __launch_bounds__(256) kc_shard_wave_init_kernel(const u64 *__restrict__ link, const u64 *__restrict__ entries, u64 n, const u64 *__restrict__ f_exit,
                                                                 const u32 *__restrict__ fidx, u64 rlo2, int me, u64 *pay, u64 *poff, u8 *flags) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool head = LK_STATE(link[entries[t]]) == LK_DEAD;
    pay[t] = head ? KC_GID(me, fidx[f_exit[t] - rlo2]) : KC_NONE64;
    poff[t] = 0;
    flags[t] = head ? 1 : 0;
}
// senders of a round = the fragments reached in the previous one that have a successor
__global__ void __launch_bounds__(256) kc_shard_wave_flag_kernel(const u8 *__restrict__ reached_now, const u64 *__restrict__ f_partner, u64 n, u8 *send) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) send[t] = reached_now[t] && f_partner[t] != KC_NONE64;
}
__global__ void __launch_bounds__(256) kc_shard_wave_send_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ f_partner, const u64 *__restrict__ f_len,
                                                                 const u64 *__restrict__ pay, const u64 *__restrict__ poff, u64 rcap2, u64 *msgs, u8 *dest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = act[t], p = f_partner[e];
    msgs[3 * t] = p; msgs[3 * t + 1] = pay[e]; msgs[3 * t + 2] = poff[e] + f_len[e];
    dest[t] = (u8)(p / rcap2);
}
__global__ void __launch_bounds__(256) kc_shard_wave_recv_kernel(const u64 *__restrict__ m, u64 n, const u32 *__restrict__ fidx, u64 rlo2, u64 *pay, u64 *poff, u8 *reached_next) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 e = fidx[m[3 * t] - rlo2];
    pay[e] = m[3 * t + 1]; poff[e] = m[3 * t + 2];
    reached_next[e] = 1;
}
// what pointer jumping would have left behind: dist = nodes from the fragment's first node to the chain's end, last = the chain's
// final fragment, nxt = none; fragments no wave reached (cross-rank cycles) keep their successor port in nxt and are flagged
__global__ void __launch_bounds__(256) kc_shard_wave_finish_kernel(u64 n, const u64 *__restrict__ f_exit, const u32 *__restrict__ fidx, u64 rlo2, const u64 *__restrict__ f_len,
                                                                   const u64 *__restrict__ f_partner, const u64 *__restrict__ pay, const u64 *__restrict__ poff, u64 *dist,
                                                                   u64 *last, u64 *nxt, u8 *unreached) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u32 tw = fidx[f_exit[t] - rlo2];
    const bool ok = pay[t] != KC_NONE64 && pay[tw] != KC_NONE64;
    dist[t] = ok ? poff[tw] + f_len[t] : f_len[t];
    last[t] = ok ? pay[tw] : KC_NONE64;
    nxt[t] = ok ? KC_NONE64 : f_partner[t];
    unreached[t] = ok ? 0 : 1;
}

// ---- slotted exchange: ONE collective per routed exchange.  kcr_build / kcr_forward need the exact count matrix before any data moves
// (a count all-gather, then the all-to-all: two collectives, three stream drains).  Here every rank sends every peer a slot of the SAME
// agreed size -- a 4-word header {items in the slot, the sender's largest per-peer count, overflow flag, items the sender has in all}
// and `cap` items -- so no count exchange is needed, and since every rank receives every rank's header, all ranks know the global
// maximum, the global item count and whether anybody overflowed without a collective of their own: an overflow is repaired by all ranks
// doubling `cap` and repeating the exchange (consumers run only after a clean exchange).  `cap` itself is derived from replicated
// values only (an all-reduced bound, or the previous exchange's global maximum).  Measured on the thread transport (8 ranks on one GPU):
// a wave round cost 0.5 ms through flag / compaction / count exchange / permutation / gather; rounds, not bytes, were what the
// ranking of the chain fragments took its time for.
#define KC_SLOT_HW 4
__global__ void __launch_bounds__(256) kc_slot_pack_kernel(const u64 *__restrict__ items, const u8 *__restrict__ dest, u64 n, int E, u64 *cnt, u64 *send, u64 cap, u64 stride,
                                                           u64 *slotpos) {
    __shared__ u32 h[KC_MAX_RANKS_SH];
    __shared__ u64 base[KC_MAX_RANKS_SH];
    const u64 tile = (u64)blockIdx.x * 2048;
    if (threadIdx.x < KC_MAX_RANKS_SH) h[threadIdx.x] = 0;
    __syncthreads();
    u32 mine[8]; u8 d[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 i = tile + (u64)j * 256 + threadIdx.x;
        d[j] = 0; mine[j] = 0;
        if (i < n) { d[j] = dest[i]; mine[j] = atomicAdd(&h[d[j]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < KC_MAX_RANKS_SH && h[threadIdx.x]) base[threadIdx.x] = atomicAdd((ull *)&cnt[threadIdx.x], (ull)h[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 i = tile + (u64)j * 256 + threadIdx.x;
        if (i >= n) continue;
        const u64 pos = base[d[j]] + mine[j];
        if (slotpos) slotpos[i] = (u64)d[j] * cap + pos;
        if (pos >= cap) continue;   // (counted all the same: the header reports the overflow)
        u64 *o = send + (u64)d[j] * stride + KC_SLOT_HW + pos * (u64)E;
        for (int w = 0; w < E; ++w) o[w] = items[i * (u64)E + w];
    }
}
__global__ void __launch_bounds__(64) kc_slot_header_kernel(u64 *cnt, int G, u64 cap, u64 *send, u64 stride) {
    const int p = threadIdx.x;
    const u64 c = p < G ? cnt[p] : 0;
    u64 mx = c, tot = c;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const u64 a = __shfl_xor(mx, o, 64), b = __shfl_xor(tot, o, 64); mx = a > mx ? a : mx; tot += b; }
    if (p < G) {
        u64 *hd = send + (u64)p * stride;
        hd[0] = c < cap ? c : cap; hd[1] = mx; hd[2] = mx > cap ? 1 : 0; hd[3] = tot;
        cnt[p] = 0;
    }
}
// out[0] = somebody overflowed, [1] = largest per-peer count anywhere, [2] = items of all ranks, [3] = items received here
__global__ void __launch_bounds__(64) kc_slot_summary_kernel(const u64 *__restrict__ recv, int G, u64 stride, u64 *out) {
    const int p = threadIdx.x;
    u64 ov = 0, mx = 0, tot = 0, here = 0;
    if (p < G) { const u64 *hd = recv + (u64)p * stride; here = hd[0]; mx = hd[1]; ov = hd[2]; tot = hd[3]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 a = __shfl_xor(mx, o, 64);
        mx = a > mx ? a : mx; ov |= __shfl_xor(ov, o, 64); tot += __shfl_xor(tot, o, 64); here += __shfl_xor(here, o, 64);
    }
    if (p == 0) { out[0] = ov; out[1] = mx; out[2] = tot; out[3] = here; }
}
// (consumers of a slotted buffer: thread t looks at item t % cap of the slot from rank t / cap)
#define KC_SLOT_ITEM(recv, stride, cap, E, t, valid, ptr)                                           \
    const u64 sl__ = (t) / (cap), it__ = (t) % (cap);                                               \
    const bool valid = it__ < (recv)[sl__ * (stride)];                                              \
    const u64 *ptr = (recv) + sl__ * (stride) + KC_SLOT_HW + it__ * (u64)(E)

// ---- sampled ranking of the chain fragments (bidirectional sets whose chains are long in fragments).  The plain waves below need as
// many rounds as the longest chain has fragments (1 kbp reads over 8 ranks: ~145 rounds, each a collective); pointer jumping needs
// log2 of that but keeps every fragment exchanging 24 bytes per round.  Here the waves stop at SPLITTERS -- chain heads and every
// fragment whose entry port hashes to 0 mod C -- so R rounds cover all gaps up to R fragments; what is still unreached after R rounds
// turns into splitters of its own (round R + 1: every receiver then is a splitter, all waves end).  Every splitter now knows the
// splitter before it and the nodes between them: that reduced list (1/C of the fragments + the few unreached ones) is ranked by
// pointer jumping TOWARDS THE HEADS (log2 of the splitters per chain rounds), and every other fragment pulls {payload, nodes before}
// of its splitter once.  The result is what the plain waves leave behind -- pay = the head's payload, poff = nodes before the
// fragment -- so kc_shard_wave_finish_kernel and everything downstream are unchanged.  Fragments on cycles across ranks end with
// pay = none, as with the waves (splitters on a cycle never reach a head: the jumping is bounded by log2 of all splitters).
// spl: 0 fragment, 1 hashed splitter, 2 unreached after R rounds, 3 chain head.
__global__ void __launch_bounds__(256) kc_shard_sr_init_kernel(const u64 *__restrict__ link, const u64 *__restrict__ entries, u64 n, const u64 *__restrict__ f_exit,
                                                               const u64 *__restrict__ f_partner, const u32 *__restrict__ fidx, u64 rlo2, int me, u32 C, u64 *pay, u64 *poff,
                                                               u64 *sjump, u8 *spl, u64 *act, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool head = LK_STATE(link[entries[t]]) == LK_DEAD;
    const bool hs = !head && kc_mix64(entries[t] ^ 0x5bd1e9955bd1e995ULL) % C == 0;
    spl[t] = head ? 3 : (hs ? 1 : 0);
    pay[t] = head ? KC_GID(me, fidx[f_exit[t] - rlo2]) : KC_NONE64;
    poff[t] = 0;
    sjump[t] = KC_NONE64;
    if ((head || hs) && f_partner[t] != KC_NONE64) act[atomicAdd((ull *)cursor, 1ULL)] = t;
}
// round R + 1: fragments no wave has reached become splitters and start waves of their own
__global__ void __launch_bounds__(256) kc_shard_sr_unreached_kernel(u64 n, const u64 *__restrict__ f_partner, const u64 *__restrict__ pay, u8 *spl, u64 *act, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || spl[t] || pay[t] != KC_NONE64) return;
    spl[t] = 2;
    if (f_partner[t] != KC_NONE64) act[atomicAdd((ull *)cursor, 1ULL)] = t;
}
// a wave message is two words: {target port (38 bits) | nodes, high 26 bits} {origin splitter (46 bits) | nodes, low 18 bits}
__global__ void __launch_bounds__(256) kc_shard_sr_send_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ f_partner, const u64 *__restrict__ f_len,
                                                               const u64 *__restrict__ pay, const u64 *__restrict__ poff, const u8 *__restrict__ spl, int me, u64 rcap2, u64 *msgs,
                                                               u8 *dest, u64 *flags) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = act[t], p = f_partner[e];
    const bool s = spl[e] != 0;
    const u64 org = s ? KC_GID(me, e) : pay[e], acc = s ? f_len[e] : poff[e] + f_len[e];
    if ((acc >> 44) || (p >> 38)) flags[0] = 1;   // (2^44 nodes between two splitters, 2^38 ports: beyond any table this library builds)
    msgs[2 * t] = p | ((acc >> 18) << 38);
    msgs[2 * t + 1] = org | ((acc & (((u64)1 << 18) - 1)) << 46);
    dest[t] = (u8)(p / rcap2);
}
__global__ void __launch_bounds__(256) kc_shard_sr_recv_kernel(const u64 *__restrict__ recv, u64 stride, u64 cap, int G, const u32 *__restrict__ fidx, u64 rlo2,
                                                               const u64 *__restrict__ f_partner, const u8 *__restrict__ spl, u64 *pay, u64 *poff, u64 *sjump, u64 *act, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)G * cap) return;
    KC_SLOT_ITEM(recv, stride, cap, 2, t, valid, m);
    if (!valid) return;
    const u64 port = m[0] & (((u64)1 << 38) - 1), org = m[1] & (((u64)1 << 46) - 1), acc = ((m[0] >> 38) << 18) | (m[1] >> 46);
    const u32 e = fidx[port - rlo2];
    poff[e] = acc;
    if (spl[e]) { sjump[e] = org; return; }   // the wave ends here: this splitter now knows the one before it and the nodes between
    pay[e] = org;
    if (f_partner[e] != KC_NONE64) act[atomicAdd((ull *)cursor, 1ULL)] = e;
}
// the splitters that have a splitter before them (everything but the heads): the reduced list, jumped towards the heads
__global__ void __launch_bounds__(256) kc_shard_sr_list_kernel(u64 n, const u8 *__restrict__ spl, const u64 *__restrict__ sjump, const u64 *__restrict__ pay, int what, u64 *act,
                                                               u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool take = what == 0 ? ((spl[t] == 1 || spl[t] == 2) && sjump[t] != KC_NONE64)   // splitters still jumping
                                : (spl[t] == 0 && pay[t] != KC_NONE64);                       // fragments that pull from their splitter
    if (take) act[atomicAdd((ull *)cursor, 1ULL)] = t;
}
__global__ void __launch_bounds__(256) kc_shard_sr_req_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ ref, u64 *req, u8 *dest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 g = ref[act[t]];
    req[t] = KC_GID_IDX(g);
    dest[t] = (u8)KC_GID_RANK(g);
}
// answers in the slots the requests came in, two words: {the splitter before the asked one, its nodes before}; an asked splitter that
// knows its head sends {payload | bit 63, nodes before} instead (ids use 46 bits; payload none: bit 62 -- a splitter nobody reached)
#define KC_SR_RES_BIT ((u64)1 << 63)
#define KC_SR_NOPAY_BIT ((u64)1 << 62)
__global__ void __launch_bounds__(256) kc_shard_sr_ans_kernel(const u64 *__restrict__ recv, u64 stride, u64 cap, int G, const u64 *__restrict__ sjump, const u64 *__restrict__ poff,
                                                              const u64 *__restrict__ pay, u64 *ans) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (u64)G * cap) return;
    KC_SLOT_ITEM(recv, stride, cap, 1, t, valid, m);
    if (!valid) return;
    const u64 u = m[0];
    u64 *a = ans + t * 2;
    a[0] = sjump[u] != KC_NONE64 ? sjump[u] : (pay[u] == KC_NONE64 ? (KC_SR_RES_BIT | KC_SR_NOPAY_BIT) : (KC_SR_RES_BIT | pay[u]));
    a[1] = poff[u];
}
__global__ void __launch_bounds__(256) kc_shard_sr_apply_jump_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ slotpos, const u64 *__restrict__ ansb, u64 *sjump,
                                                                     u64 *poff, u64 *pay, u64 *act_next, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = act[t];
    const u64 *a = ansb + slotpos[t] * 2;
    poff[e] += a[1];
    if (a[0] & KC_SR_RES_BIT) { pay[e] = (a[0] & KC_SR_NOPAY_BIT) ? KC_NONE64 : (a[0] & ~KC_SR_RES_BIT); sjump[e] = KC_NONE64; }
    else { sjump[e] = a[0]; act_next[atomicAdd((ull *)cursor, 1ULL)] = e; }
}
__global__ void __launch_bounds__(256) kc_shard_sr_apply_pull_kernel(const u64 *__restrict__ act, u64 n, const u64 *__restrict__ slotpos, const u64 *__restrict__ ansb, u64 *poff,
                                                                     u64 *pay) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = act[t];
    const u64 *a = ansb + slotpos[t] * 2;
    // (a splitter on a cycle never found a head: it is still jumping or was given up -- no payload for the fragments behind it either)
    const bool ok = (a[0] & KC_SR_RES_BIT) && !(a[0] & KC_SR_NOPAY_BIT);
    pay[e] = ok ? (a[0] & ~KC_SR_RES_BIT) : KC_NONE64;
    poff[e] += a[1];
}
// splitters that are still jumping when the bound is reached lie on cycles
__global__ void __launch_bounds__(256) kc_shard_sr_giveup_kernel(const u64 *__restrict__ act, u64 n, u64 *pay) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) pay[act[t]] = KC_NONE64;
}

// ---- the branch subgraph settled on the DEVICE (round 5).  Every rank holds all its edges (all-gathered rows {x pid, y pid, priority,
// base, key x, key y}: every edge of every branch port, an edge between two branch ports twice) and settles them identically: the
// greedy matching by descending priority of the canonical rule, computed as the fixpoint of "an edge is taken when it is the best
// still-available edge at BOTH its ports" (the same matching, whatever the order things happen in).  The host version (settle_branches:
// std::sort, hash maps) is exact but took 12 s for the 1.2 10^7 edges of 10x reads with 1 % errors -- 50x the whole single-GPU run --
// and stays as the fallback for the one thing this does not decide itself: two different edges with the same 64-bit priority.
__device__ __forceinline__ u64 kc_lower_bound_u64(const u64 *__restrict__ a, u64 n, u64 v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void __launch_bounds__(256) kc_br_keys_kernel(const u64 *__restrict__ rows, u64 n, int EW, u64 *key, u64 *idx) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { key[t] = ~rows[t * (u64)EW + 2]; idx[t] = t; }   // ascending ~priority = descending priority
}
// sorted position j holds row perm[j]: keep[j] = a new edge (not the second listing of the one before, not a port with itself);
// flags[0] = two DIFFERENT edges share a priority (the host settles the subgraph then)
__global__ void __launch_bounds__(256) kc_br_mark_kernel(const u64 *__restrict__ rows, const u64 *__restrict__ perm, u64 n, int EW, u8 *keep, u64 *flags) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u64 *e = rows + perm[j] * (u64)EW;
    bool k = e[0] != e[1];
    // every earlier row of the same priority is looked at (a run is one or two rows unless 64-bit priorities collide): the second
    // listing of an edge need not sit right behind the first when a third row -- a self-loop, another edge -- shares the priority
    for (u64 i = j; i > 0; --i) {
        const u64 *p = rows + perm[i - 1] * (u64)EW;
        if (p[2] != e[2]) break;
        const bool same = (p[0] == e[1] && p[1] == e[0]) || (p[0] == e[0] && p[1] == e[1]);
        if (same) k = false; else if (e[0] != e[1] && p[0] != p[1]) flags[0] = 1;
    }
    keep[j] = k ? 1 : 0;
}
__global__ void __launch_bounds__(256) kc_br_ports_kernel(const u64 *__restrict__ rows, const u64 *__restrict__ uedge, u64 ne, int EW, u64 *ports) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ne) return;
    const u64 *e = rows + uedge[r] * (u64)EW;
    ports[2 * r] = e[0]; ports[2 * r + 1] = e[1];
}
__global__ void __launch_bounds__(256) kc_br_uniq_flags_kernel(const u64 *__restrict__ a, u64 n, u8 *f) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) f[t] = (t == 0 || a[t] != a[t - 1]) ? 1 : 0;
}
// adjacency entries of edge r (rank by priority): {port index << 32 | r} -> the port at its other end
__global__ void __launch_bounds__(256) kc_br_adj_kernel(const u64 *__restrict__ rows, const u64 *__restrict__ uedge, u64 ne, int EW, const u64 *__restrict__ uports, u64 np,
                                                        u64 *akey, u64 *aval) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ne) return;
    const u64 *e = rows + uedge[r] * (u64)EW;
    const u64 a = kc_lower_bound_u64(uports, np, e[0]), b = kc_lower_bound_u64(uports, np, e[1]);
    akey[2 * r] = (a << 32) | r; aval[2 * r] = b;
    akey[2 * r + 1] = (b << 32) | r; aval[2 * r + 1] = a;
}
__global__ void __launch_bounds__(256) kc_br_branchflag_kernel(const u64 *__restrict__ rows, u64 n, int EW, const u64 *__restrict__ uports, u64 np, u8 *isbranch) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 x = rows[t * (u64)EW];
    const u64 a = kc_lower_bound_u64(uports, np, x);
    if (a < np && uports[a] == x) isbranch[a] = 1;
}
// every unmatched port picks its best edge whose other end is unmatched too (the adjacency of a port is sorted by rank)
__global__ void __launch_bounds__(256) kc_br_pick_kernel(const u64 *__restrict__ akey, const u64 *__restrict__ aval, u64 na, u64 np, const u32 *__restrict__ mate, u32 *pick) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    u32 best = 0xffffffffu;
    if (mate[p] == 0xffffffffu)
        for (u64 i = kc_lower_bound_u64(akey, na, p << 32); i < na && (akey[i] >> 32) == p; ++i)
            if (mate[aval[i]] == 0xffffffffu) { best = (u32)akey[i]; break; }
    pick[p] = best;
}
__global__ void __launch_bounds__(256) kc_br_accept_kernel(const u64 *__restrict__ akey, const u64 *__restrict__ aval, u64 na, u64 np, const u32 *__restrict__ pick, u32 *mate, u64 *changed) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    const u32 r = pick[p];
    if (r == 0xffffffffu) return;
    const u64 i = kc_lower_bound_u64(akey, na, (p << 32) | (u64)r);   // the entry of (p, r): its value is the other end
    const u64 q = aval[i];
    if (pick[q] == r) { mate[p] = (u32)q; if (p < q) atomicAdd((ull *)changed, 1ULL); }
}
// the link words that change, as the host version lists them: a branch port takes its matched edge or dies; a port with this one
// candidate only keeps its choice if the edge was taken and dies otherwise
__global__ void __launch_bounds__(256) kc_br_emit_kernel(const u64 *__restrict__ rows, u64 n, int EW, const u64 *__restrict__ uports, u64 np, const u32 *__restrict__ mate,
                                                         const u8 *__restrict__ isbranch, u32 *emitted, u64 *fix, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 *e = rows + t * (u64)EW;
    const u64 x = e[0], y = e[1];
    const u64 a = kc_lower_bound_u64(uports, np, x), b = kc_lower_bound_u64(uports, np, y);
    const bool ha = a < np && uports[a] == x, hb = b < np && uports[b] == y;   // (a port whose only rows were edges to itself is in no edge)
    const u32 ma = ha ? mate[a] : 0xffffffffu;
    if (ma == 0xffffffffu) {
        // (once per port: its rows follow each other, but a flag costs nothing and does not depend on that)
        const bool first = ha ? atomicExch(&emitted[a], 1u) == 0u : (t == 0 || rows[(t - 1) * (u64)EW] != x);
        if (first) { const u64 pos = atomicAdd((ull *)cursor, 1ULL); fix[2 * pos] = x; fix[2 * pos + 1] = LK_DEAD; }
    } else if (hb && ma == (u32)b) {
        const u64 pos = atomicAdd((ull *)cursor, 1ULL);
        fix[2 * pos] = x; fix[2 * pos + 1] = LK_MAKE(LK_MATCHED, y & 1, e[3], y >> 1);
    }
    if (hb && !isbranch[b] && mate[b] == 0xffffffffu && atomicExch(&emitted[b], 1u) == 0u) {
        const u64 pos = atomicAdd((ull *)cursor, 1ULL);
        fix[2 * pos] = y; fix[2 * pos + 1] = LK_DEAD;
    }
}

// ---- cycles.  K-mers that no fragment reaches lie on cycles inside this rank, fragments that the ranking never connects to a
// chain end lie on cycles that span ranks.  Both kinds are dealt with together after the ranking: the rows of their k-mers are
// all-gathered (they are few), every rank looks up the candidate edges of the ports whose home it owns (kc_shard_cycle_cand_kernel),
// those are all-gathered too, and every rank takes the same decisions as the single-GPU flow and the oracle (kc_cycles_decide).
__global__ void __launch_bounds__(256) kc_shard_sum_kernel(const u64 *__restrict__ a, u64 n, u64 *out) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += a[i];
    u64 tot;
    (void)kc_block_excl_scan(s, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd((ull *)out, (ull)tot);
}
template <bool FILL>
__global__ void __launch_bounds__(256) kc_shard_unvisited_kernel(const u32 *__restrict__ ncount, u32 nstride, const u8 *__restrict__ visited, u64 rlo, u64 rcap, u64 *list, u64 *cursor) {
    const u64 t = kc_gtid();
    if (t >= rcap || (u32)((rlo + t) % nstride) >= ncount[(rlo + t) / nstride] || visited[t]) return;
    const u64 pos = atomicAdd((ull *)cursor, 1ULL);
    if (FILL) list[pos] = rlo + t;
}
// mark the k-mers of the fragments that are on chains with ends (skip[t]: fragment t was never reached by the ranking)
__global__ void __launch_bounds__(256) kc_shard_mark_kernel(const u64 *__restrict__ link, u64 rlo, u64 rcap, const u64 *__restrict__ entries, u64 n, const u8 *__restrict__ skip, u8 *visited) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || skip[t]) return;
    const u64 h = entries[t];
    u64 cur = h >> 1;
    int co = (int)(h & 1);
    while (true) {
        visited[cur - rlo] = 1;
        const u64 w = link[(cur << 1) | (u64)(co == 0 ? 1 : 0)];
        if (!LK_IS_MATCHED(w) || LK_SLOT(w) - rlo >= rcap) break;
        cur = LK_SLOT(w);
        co = LK_PORT(w) == 0 ? 0 : 1;
    }
}
// candidate edges of the ports of the gathered cycle k-mers (rows: {slot, link L, link R, key}) whose home this rank owns, to
// ports that the matching left dead (rows as kc_cycle_cand_kernel writes them)
template <int W>
__global__ void __launch_bounds__(256) kc_shard_cycle_cand_kernel(SetView<W> set, const u64 *__restrict__ link, int k, int complements, const u64 *__restrict__ rows, u64 n,
                                                                  u64 *cand, u64 cand_cap, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const u64 *r = rows + (t >> 1) * (3 + W);
    const int port = (int)(t & 1);
    Key<W> x;
    for (int w = 0; w < W; ++w) x.w[w] = r[3 + w];
    const Key<W> xr = key_rc(x, k);
    const Key<W> lm = key_lowmask<W>(2 * (k - 2));
    const Key<W> mid = port ? key_and(x, lm) : key_shr(x, 4), midr = port ? key_shr(xr, 4) : key_and(xr, lm);
    const KcPortDesc d = kc_port_desc<W, 0>(set, x, xr, k, complements, port, kc_mid_min<W, 0>(mid, midr, k, set.m));
    if (d.home - set.rlo >= set.rcap) return;
    u64 cs[4]; u32 qb;
    kc_port_search_generic<W>(set, k, complements, x, xr, port, cs, qb);
    const u64 hx = key_hash(x);
    for (u32 c = 0; c < 4; ++c) {
        u64 ys = KC_NONE64;
#pragma unroll
        for (u32 cc = 0; cc < 4; ++cc) ys = (c == cc) ? cs[cc] : ys;
        if (ys == KC_NONE64) continue;
        ys = set_nid_of_slot(set, ys);   // (ports are numbered by dense node id)
        const u64 yw = link[(ys << 1) | (u64)((qb >> c) & 1u)];
        if (LK_STATE(yw) != LK_DEAD || (yw & LK_CUT)) continue;
        int q;
        const Key<W> y = kc_neighbour_key_q(x, xr, port, c, k, complements, q);
        kc_cycle_cand_emit<W>(x, hx, r[0], port, y, ys, q, k, cand, cand_cap, cursor);
    }
}
// ---- chain heads: fragments whose entry port is dead.  They ask the owner of the chain's final fragment for the tail
// (oriented tail node + its key) and decide whether their direction is the one that is emitted.
__global__ void __launch_bounds__(256) kc_shard_head_flags_kernel(const u64 *__restrict__ link, const u64 *__restrict__ entries, u64 n, u8 *flags) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) flags[t] = LK_STATE(link[entries[t]]) == LK_DEAD;
}
__global__ void __launch_bounds__(256) kc_shard_tailreq_kernel(const u64 *__restrict__ heads, u64 n, const u64 *__restrict__ last, u64 *req, u8 *dest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 g = last[heads[t]];
    req[t] = g; dest[t] = (u8)KC_GID_RANK(g);
}
template <int W>
__global__ void __launch_bounds__(256) kc_shard_tailans_kernel(SetView<W> set, const u64 *__restrict__ link, const u64 *__restrict__ req, u64 n, const u64 *__restrict__ f_exit, u64 *ans) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 x = f_exit[KC_GID_IDX(req[t])];
    ans[t * (W + 1)] = x;
    const Key<W> key = set_key_at(set, LK_SLOT(link[x]));   // (the chain ends on a dead port: its word carries the slot of the k-mer)
    for (int w = 0; w < W; ++w) ans[t * (W + 1) + 1 + w] = key.w[w];
}
// chain record (3 + 3W words): {final fragment id, nodes, head key (canonical), first k-mer (oriented), last k-mer (oriented)}
template <int W>
__global__ void __launch_bounds__(256) kc_shard_chain_kernel(SetView<W> set, const u64 *__restrict__ link, int k, int complements, const u64 *__restrict__ heads, u64 n,
                                                             const u64 *__restrict__ entries, const u64 *__restrict__ dist, const u64 *__restrict__ last,
                                                             const u64 *__restrict__ tail, u64 *chains, u64 *cursor) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 e = heads[t], h = entries[e], len = dist[e];
    const Key<W> hk = set_key_at(set, LK_SLOT(link[h]));   // (a head's entry port is dead: its word carries the slot)
    Key<W> tk;
    for (int w = 0; w < W; ++w) tk.w[w] = tail[t * (W + 1) + 1 + w];
    const u64 tx = tail[t * (W + 1)];   // tail node << 1 | exit port
    bool winner = true;
    if (complements) winner = (len == 1) ? ((h & 1) == 0) : key_lt(hk, tk);
    atomicAdd((ull *)(cursor + 1), (ull)(winner ? len : 0));
    if (!winner) return;
    const u64 pos = atomicAdd((ull *)cursor, 1ULL);
    u64 *c = chains + pos * (3 + 3 * W);
    c[0] = last[e]; c[1] = len; c[2] = 0;
    const Key<W> fk = (h & 1) ? key_rc(hk, k) : hk;
    const Key<W> lk = (tx & 1) ? tk : key_rc(tk, k);   // orientation of the tail node = 1 - exit port; orientation 1 = reverse complement
    for (int w = 0; w < W; ++w) { c[3 + w] = hk.w[w]; c[3 + W + w] = fk.w[w]; c[3 + 2 * W + w] = lk.w[w]; }
}

// ---- the replicated simplitig list: chain records sorted by head key (device radix sort over the key words) -> lens / word
// counts / firstK / lastK / final fragment ids; the final-fragment ids sorted once more for the fragments' membership test
__global__ void __launch_bounds__(256) kc_shard_row_keys_kernel(const u64 *__restrict__ rows, u64 n, int CW, int W, u64 *out) {   // the head keys of the chain records, n x W words
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) for (int w = 0; w < W; ++w) out[t * (u64)W + w] = rows[t * (u64)CW + 3 + w];
}
__global__ void __launch_bounds__(256) kc_shard_simp_fill_kernel(const u64 *__restrict__ rows, const u64 *__restrict__ perm, u64 S, int CW, int W, int k, u64 *lens,
                                                                 u64 *nwords, u64 *firstK, u64 *lastK, u64 *fin, u64 *idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const u64 *r = rows + perm[i] * (u64)CW;
    const u64 L = r[1] + (u64)(k - 1);
    lens[i] = L; nwords[i] = L / 32 + 1;
    for (int w = 0; w < W; ++w) { firstK[i * W + w] = r[3 + W + w]; lastK[i * W + w] = r[3 + 2 * W + w]; }
    fin[i] = r[0]; idx[i] = i;
}
__device__ __forceinline__ u64 kc_lower_bound64(const u64 *a, u64 n, u64 v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// ---------------------------------------------------------------------------------------------------- phase 3: bases to the emitting rank
// fragment e of the emitted direction of simplitig i covers string positions [p0, p0 + cnt): p0 = 0, cnt = k - 1 + len for
// the head fragment, p0 = node offset + k - 1, cnt = len otherwise.  One message per 32 bases:
// {absolute base position in the packed stream | (count - 1) << 58, 32 bases left-aligned}.
template <bool FILL>
__global__ void __launch_bounds__(256) kc_shard_msg_kernel(int k, const u64 *__restrict__ entries, u64 n, const u64 *__restrict__ f_len, const u64 *__restrict__ dist,
                                                           const u64 *__restrict__ last, const u32 *__restrict__ f_stream, const u64 *__restrict__ pool,
                                                           const u32 *__restrict__ chunk_next, const u64 *__restrict__ keys_biased, const u64 *__restrict__ link, u32 nstride, int bshift, int W,
                                                           const u64 *__restrict__ fin_sorted,
                                                           const u64 *__restrict__ fin_simp, u64 S, const u64 *__restrict__ lens, const u64 *__restrict__ woff,
                                                           const u8 *__restrict__ simp_dest, u64 *mcount, const u64 *__restrict__ moff, u64 *msgs, u8 *mdest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 fin = last[t];
    const u64 j = kc_lower_bound64(fin_sorted, S, fin);
    if (j >= S || fin_sorted[j] != fin) { if (!FILL) mcount[t] = 0; return; }   // the other direction is the emitted one
    const u64 i = fin_simp[j];
    const u64 nodes = lens[i] - (u64)(k - 1);
    const u64 off = nodes - dist[t], len = f_len[t];
    const bool head = off == 0;
    const u64 cnt = head ? (u64)(k - 1) + len : len;
    const u64 nm = (cnt + 31) / 32;
    if (!FILL) { mcount[t] = nm; return; }
    const u64 p0 = head ? 0 : off + (u64)(k - 1);
    // oriented first k-mer of the fragment
    const u64 h = entries[t];
    u64 kw[4] = {0, 0, 0, 0};
    // the fragment's first k-mer: the word of its entry port (dead, or matched to a node of another rank) carries the slot offset
    const u64 hslot = (((h >> 1) / nstride) << bshift) | LK_OWN(link[h]);
    for (int w = 0; w < W; ++w) kw[w] = keys_biased[hslot * (u64)W + w];
    auto base_at = [&](int idx) -> u64 {   // idx-th base of the oriented k-mer
        if (h & 1) {   // reverse complement: base idx = complement of forward base k - 1 - idx
            const int s = 2 * idx;
            return 3 - ((kw[s >> 6] >> (s & 63)) & 3);
        }
        const int s = 2 * (k - 1 - idx);
        return (kw[s >> 6] >> (s & 63)) & 3;
    };
    u64 m = moff[t], pos = woff[i] * 32 + p0, acc = 0;
    int nacc = 0;
    const u8 d = simp_dest[i];
    auto put = [&](u64 b) {
        acc = (acc << 2) | b;
        if (++nacc == 32) { msgs[2 * m] = pos | ((u64)31 << 58); msgs[2 * m + 1] = acc; mdest[m] = d; ++m; pos += 32; acc = 0; nacc = 0; }
    };
    if (head) for (int idx = 0; idx < k; ++idx) put(base_at(idx));
    else put(base_at(k - 1));
    u32 chunk = f_stream[t];
    int wi = 0;
    u64 left = len - 1;
    while (left) {
        if (wi == KC_FCW) { chunk = chunk_next[chunk]; wi = 0; }
        const u64 sw = pool[(u64)chunk * KC_FCW + (u64)wi++];
        const int nb = left >= 32 ? 32 : (int)left;
        for (int q = 0; q < nb; ++q) put((sw >> (62 - 2 * q)) & 3);
        left -= (u64)nb;
    }
    if (nacc) { msgs[2 * m] = pos | ((u64)(nacc - 1) << 58); msgs[2 * m + 1] = acc << (64 - 2 * nacc); mdest[m] = d; }
}
__global__ void __launch_bounds__(256) kc_shard_place_kernel(const u64 *__restrict__ msgs, u64 n, u64 *words) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const u64 hdr = msgs[2 * t], payload = msgs[2 * t + 1];
    const u64 pos = hdr & (((u64)1 << 58) - 1);
    const int cnt = (int)(hdr >> 58) + 1;
    const u64 wi = pos >> 5;
    const int sh = (int)(pos & 31) * 2;   // bits already used in word wi
    atomicOr((ull *)&words[wi], (ull)(payload >> sh));
    if (sh && sh + 2 * cnt > 64) atomicOr((ull *)&words[wi + 1], (ull)(payload << (64 - sh)));
}

// ---- -M (global.h:185-195) on the sharded path: every lower-case position of this rank's slice asks the OWNER of the k-mer that
// starts there whether the set holds it (the k-mer may run up to k-1 characters into the following slices: `halo`).
// pass 1 (FILL = false): mask = out, lower-case k-mer starts are counted per thread tile; pass 2: queries {key} + position.
template <int W, bool FILL>
__global__ void __launch_bounds__(256) kc_shard_mask_query_kernel(const char *__restrict__ out, u64 n, const char *__restrict__ halo, u64 goff, u64 L, int k, int complements,
                                                                  SetView<W> set, u64 bpr_slots, char *mask, u64 *tile_count, const u64 *__restrict__ tile_off, u64 *qkeys,
                                                                  u64 *qpos, u8 *qdest) {
    const u64 p = kc_gtid();
    bool ask = false;
    Key<W> x = key_zero<W>();
    if (p < n) {
        const char c = out[p];
        if (!FILL) mask[p] = c;
        if (c > 'Z' && goff + p + (u64)k <= L) {
            ask = true;
            if (FILL) {
                for (int i = 0; i < k; ++i) {
                    const u64 q = p + (u64)i;
                    const u32 up = ((u32)(u8)(q < n ? out[q] : halo[q - n])) & 0xDFu;
                    u32 code = (up >> 1) & 3u;
                    code ^= code >> 1;
                    x = key_append(x, (u64)code, k);
                }
                if (complements) { const Key<W> r = key_rc(x, k); if (key_lt(r, x)) x = r; }
            }
        }
    }
    u64 tot;
    const u64 ex = kc_block_excl_scan(ask ? 1 : 0, &tot);
    if (!FILL) { if (threadIdx.x == 0) tile_count[(u64)blockIdx.y * gridDim.x + blockIdx.x] = tot; return; }
    if (ask) {
        const u64 at = tile_off[(u64)blockIdx.y * gridDim.x + blockIdx.x] + ex;
        for (int w = 0; w < W; ++w) qkeys[at * W + w] = x.w[w];
        qpos[at] = p;
        qdest[at] = (u8)(set_place(set, x).home / bpr_slots);
    }
}
template <int W>
__global__ void __launch_bounds__(256) kc_shard_mask_answer_kernel(SetView<W> set, const u64 *__restrict__ q, u64 n, u64 *ans) {
    const u64 t = kc_gtid();
    if (t >= n) return;
    Key<W> key;
    for (int w = 0; w < W; ++w) key.w[w] = q[t * W + w];
    ans[t] = set_find(set, key) != KC_NONE64 ? 1 : 0;
}
__global__ void __launch_bounds__(256) kc_shard_mask_apply_kernel(const u64 *__restrict__ qpos, const u64 *__restrict__ ans, u64 n, char *mask) {
    const u64 t = kc_gtid();
    if (t < n && ans[t]) mask[qpos[t]] = (char)(mask[qpos[t]] - ('a' - 'A'));
}

// per-simplitig destination rank: owner of the output offset at which its on-path vertex starts
__global__ void __launch_bounds__(256) kc_shard_simp_dest_kernel(const u64 *__restrict__ dist, const u64 *__restrict__ endv, u64 start, u64 S, u64 V, u64 Ltot, int G,
                                                                 u8 *simp_dest, u64 *rank_lo, u64 *rank_hi, const u64 *__restrict__ weight) {
    // the ranks' output ranges are reduced per workgroup in LDS first: one atomicMin / atomicMax per vertex on G global words
    // serialised at ~6 ns each (12 ms per 10^6 simplitigs, on every rank)
    __shared__ u64 lo_s[64], hi_s[64];   // (at most 64 ranks: KC_MAX_RANKS of the routing code)
    for (int i = threadIdx.x; i < G; i += blockDim.x) { lo_s[i] = ~(u64)0; hi_s[i] = 0; }
    __syncthreads();
    const u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V && endv[v] == endv[start]) {
        const u64 off = dist[start] - dist[v];
        const int r = (int)(((unsigned __int128)off * (unsigned)G) / Ltot);
        simp_dest[v < S ? v : v - S] = (u8)r;
        atomicMin((ull *)&lo_s[r], (ull)off);
        atomicMax((ull *)&hi_s[r], (ull)(off + weight[v]));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < G; i += blockDim.x)
        if (hi_s[i]) { atomicMin((ull *)&rank_lo[i], (ull)lo_s[i]); atomicMax((ull *)&rank_hi[i], (ull)hi_s[i]); }
}


// ---- -S (simplitigs_from_fasta, simplitigs.h:89-103) on the sharded path: the records ARE the simplitigs.  Every rank packs its own,
// the ranks all-gather {length, first k-mer, last k-mer} per simplitig in record order (= the reference's order), phases 2-3 run as
// for computed simplitigs, and the packed words of a simplitig travel to the rank that emits it.
// rows {length, first k-mer[W], last k-mer[W]} of this rank's simplitigs
__global__ void __launch_bounds__(256) kc_shard_srows_kernel(const u64 *__restrict__ lens, const u64 *__restrict__ firstK, const u64 *__restrict__ lastK, u64 n, int W, u64 *rows) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u64 *r = rows + t * (u64)(1 + 2 * W);
    r[0] = lens[t];
    for (int w = 0; w < W; ++w) { r[1 + w] = firstK[t * W + w]; r[1 + W + w] = lastK[t * W + w]; }
}
// the replicated list from the gathered rows; nw[i] = words of simplitig i
__global__ void __launch_bounds__(256) kc_shard_sfill_kernel(const u64 *__restrict__ rows, u64 S, int W, u64 *lens, u64 *nw, u64 *firstK, u64 *lastK) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const u64 *r = rows + t * (u64)(1 + 2 * W);
    lens[t] = r[0]; nw[t] = r[0] / 32 + 1;
    for (int w = 0; w < W; ++w) { firstK[t * W + w] = r[1 + w]; lastK[t * W + w] = r[1 + W + w]; }
}
// one message per packed word of this rank's simplitigs: {position in bases | 31 << 58, the word} to the rank that emits the simplitig
__global__ void __launch_bounds__(256) kc_shard_smsg_kernel(const u64 *__restrict__ lwords, const u64 *__restrict__ lwoff, u64 n_local, u64 nwords_local, u64 g0,
                                                            const u64 *__restrict__ gwoff, const u8 *__restrict__ simp_dest, u64 *msgs, u8 *mdest) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nwords_local) return;
    u64 lo = 0, hi = n_local + 1;   // upper bound of t in lwoff[0 .. n_local]
    while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (lwoff[mid] <= t) lo = mid + 1; else hi = mid; }
    const u64 i = lo - 1, j = t - lwoff[i], g = g0 + i;
    msgs[2 * t] = ((gwoff[g] + j) * 32) | ((u64)31 << 58);
    msgs[2 * t + 1] = lwords[t];
    mdest[t] = simp_dest[g];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- host flow
namespace {

int read_words(kc_ctx *ctx, const u64 *d, u64 *h, int n) {
    KC_HIP(ctx, hipMemcpyAsync(h, d, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}
float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ms;
}
bool trace_on() { return getenv("KC_SHARD_TRACE") != nullptr; }

// every rank's `n_local` rows of `words` words -> all rows on every rank (device), row counts per rank in `counts`
int allgather_rows(kc_comm *c, const u64 *d_rows, u64 n_local, int words, u64 **d_all, u64 *n_all, std::vector<u64> *counts) {
    kc_ctx *ctx = c->ctx;
    const int G = c->world;
    std::vector<u64> cnt(G);
    KC_TRY(kcc_allgather_words(c, &n_local, cnt.data(), 1));
    std::vector<u64> so(G, 0), sc(G, n_local * words * 8), ro(G), rcv(G);
    u64 tot = 0;
    for (int p = 0; p < G; ++p) { ro[p] = tot * words * 8; rcv[p] = cnt[p] * words * 8; tot += cnt[p]; }
    u64 *all = nullptr;
    KC_TRY(kc_alloc(ctx, &all, tot * words + 1));
    int rc = kcc_alltoallv(c, d_rows, so.data(), sc.data(), all, ro.data(), rcv.data());
    if (rc != KC_OK) { kc_release(ctx, all); return rc; }
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *d_all = all; *n_all = tot;
    if (counts) *counts = cnt;
    return KC_OK;
}

// Host side of the slotted exchange (see kc_slot_pack_kernel): buffers for G slots of `cap` items, grown when an exchange overflows.
struct SlotX {
    kc_ctx *ctx; kc_comm *c; KcScope *scope;
    int G, E = 0;
    u64 cap = 0, stride = 0;
    u64 *send = nullptr, *recv = nullptr, *cnt = nullptr;   // cnt: G counters, then 4 summary words
    u64 exchanges = 0, redone = 0;
    u64 sent_to[KC_MAX_RANKS_SH] = {0}, got_from[KC_MAX_RANKS_SH] = {0};   // items per peer of the last exchange
    // KC_SLOT_TEST_SHRINK=n (tests; read when the exchange object is set up, never on the data path): every requested slot size is divided
    // by n, so that exchanges overflow and the agreed repeat with the larger size runs
    u64 test_shrink = 1;
    SlotX(kc_ctx *x, kc_comm *cm, KcScope *sc) : ctx(x), c(cm), scope(sc), G(cm->world) {
        if (const char *e = getenv("KC_SLOT_TEST_SHRINK")) { const long v = atol(e); if (v >= 1) test_shrink = (u64)v; }
    }
    int ensure(int e, u64 want_cap, bool after_overflow = false) {
        if (!after_overflow) want_cap /= test_shrink;
        if (!cnt) { KC_TRY(scope->alloc(&cnt, KC_MAX_RANKS_SH + 8)); KC_HIP(ctx, hipMemsetAsync(cnt, 0, (KC_MAX_RANKS_SH + 8) * 8, ctx->stream)); }
        if (want_cap < (test_shrink > 1 ? 1 : 64)) want_cap = test_shrink > 1 ? 1 : 64;
        const u64 st = KC_SLOT_HW + want_cap * (u64)e;
        if (send && st * (u64)G <= words_) { E = e; cap = want_cap; stride = st; return KC_OK; }
        if (send) { KC_HIP(ctx, hipStreamSynchronize(ctx->stream)); scope->release(send); scope->release(recv); send = recv = nullptr; }
        words_ = st * (u64)G;
        KC_TRY(scope->alloc(&send, words_ + 1)); KC_TRY(scope->alloc(&recv, words_ + 1));
        E = e; cap = want_cap; stride = st;
        return KC_OK;
    }
    // items: n rows of E words, row i to rank dest[i].  On return recv holds the G slots; h4 = {0, largest per-peer count anywhere, items
    // of all ranks, items received here}.  An overflow anywhere makes every rank double cap and repeat (all see the same headers).
    int exchange(const u64 *items, const u8 *dest, u64 n, u64 *slotpos, u64 h4[4]) {
        for (int attempt = 0;; ++attempt) {
            if (attempt == 24) return kc_fail(ctx, KC_E_CAPACITY, "slotted exchange: the slots keep overflowing");
            if (n) KC_LAUNCH(kc_slot_pack_kernel, dim3(kc_grid(n, 2048)), dim3(256), 0, ctx->stream, items, dest, n, E, cnt, send, cap, stride, slotpos);
            KC_LAUNCH(kc_slot_header_kernel, dim3(1), dim3(64), 0, ctx->stream, cnt, G, cap, send, stride);
            std::vector<u64> off(G), sz(G, stride * 8);
            for (int p = 0; p < G; ++p) off[p] = (u64)p * stride * 8;
            KC_TRY(kcc_alltoallv(c, send, off.data(), sz.data(), recv, off.data(), sz.data()));
            KC_LAUNCH(kc_slot_summary_kernel, dim3(1), dim3(64), 0, ctx->stream, (const u64 *)recv, G, stride, cnt + KC_MAX_RANKS_SH);
            KC_HIP(ctx, hipGetLastError());
            KC_HIP(ctx, hipMemcpyAsync(h4, cnt + KC_MAX_RANKS_SH, 32, hipMemcpyDeviceToHost, ctx->stream));
            // (how many items went to / came from every peer: the answers travel back in exactly those numbers)
            KC_HIP(ctx, hipMemcpy2DAsync(sent_to, 8, send, stride * 8, 8, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipMemcpy2DAsync(got_from, 8, recv, stride * 8, 8, (size_t)G, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ++exchanges;
            if (!h4[0]) return KC_OK;
            ++redone;
            KC_TRY(ensure(E, h4[1] + h4[1] / 4 + 64, true));   // (the largest count is known to everybody: the repeat fits)
        }
    }
    // answers of A words in the slots the requests arrived in (ans: G * cap rows) -> back[slotpos[i]] on the asking side
    int answer(const u64 *ans, int A, u64 *back) {
        std::vector<u64> off(G), ssz(G), rsz(G);
        for (int p = 0; p < G; ++p) { off[p] = (u64)p * cap * (u64)A * 8; ssz[p] = got_from[p] * (u64)A * 8; rsz[p] = sent_to[p] * (u64)A * 8; }
        KC_TRY(kcc_alltoallv(c, ans, off.data(), ssz.data(), back, off.data(), rsz.data()));
        ++exchanges;
        return KC_OK;
    }
private:
    u64 words_ = 0;
};

template <int W> bool hkey_lt(const u64 *a, const u64 *b) {
    for (int i = W - 1; i >= 0; --i) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; }
    return false;
}
template <int W> bool hkey_eq(const u64 *a, const u64 *b) {
    for (int i = 0; i < W; ++i) if (a[i] != b[i]) return false;
    return true;
}

// The branch subgraph, settled the way the canonical rule is defined (oracle restatement: simplitigs_canonical_order):
// all its edges by descending priority (ties: ordered endpoint tuple), an edge is taken when both ports are free.
// rows: {x pid, y pid, priority, base, key x, key y} as listed by kc_shard_resolve_kernel (every edge of a branch port; an
// edge between two branch ports appears twice).  fix: {pid, final link word} for every port whose provisional link changes.
template <int W> void settle_branches(const std::vector<u64> &rows, u64 n, std::vector<u64> &fix) {
    const int EW = 4 + 2 * W;
    struct Edge { u64 pa, pb, pr; const u64 *ka, *kb; };
    std::vector<Edge> edges;
    edges.reserve(n);
    std::unordered_set<u64> branch;
    for (u64 i = 0; i < n; ++i) {
        const u64 *e = rows.data() + i * EW;
        const u64 x = e[0], y = e[1];
        const u64 *kx = e + 4, *ky = e + 4 + W;
        branch.insert(x);
        const bool x_first = hkey_lt<W>(kx, ky) || (hkey_eq<W>(kx, ky) && (x & 1) <= (y & 1));
        if (x_first) edges.push_back({x, y, e[2], kx, ky}); else edges.push_back({y, x, e[2], ky, kx});
    }
    std::sort(edges.begin(), edges.end(), [](const Edge &a, const Edge &b) {
        if (a.pr != b.pr) return a.pr > b.pr;
        if (!hkey_eq<W>(a.ka, b.ka)) return hkey_lt<W>(a.ka, b.ka);
        if ((a.pa & 1) != (b.pa & 1)) return (a.pa & 1) < (b.pa & 1);
        if (!hkey_eq<W>(a.kb, b.kb)) return hkey_lt<W>(a.kb, b.kb);
        return (a.pb & 1) < (b.pb & 1);
    });
    std::unordered_map<u64, u64> mate;
    mate.reserve(edges.size() * 2);
    for (const Edge &e : edges) {
        if (e.pa == e.pb) continue;
        if (mate.count(e.pa) || mate.count(e.pb)) continue;
        mate[e.pa] = e.pb; mate[e.pb] = e.pa;
    }
    fix.clear();
    std::unordered_set<u64> done;
    for (u64 i = 0; i < n; ++i) {
        const u64 *e = rows.data() + i * EW;
        const u64 x = e[0], y = e[1];
        auto mx = mate.find(x);
        if (mx == mate.end()) { if (done.insert(x).second) { fix.push_back(x); fix.push_back(LK_DEAD); } }
        else if (mx->second == y) { fix.push_back(x); fix.push_back(LK_MAKE(LK_MATCHED, y & 1, e[3], y >> 1)); }
        if (!branch.count(y)) {   // y has this one candidate only: it keeps its choice iff the edge was taken
            auto my = mate.find(y);
            if (my == mate.end() && done.insert(y).second) { fix.push_back(y); fix.push_back(LK_DEAD); }
        }
    }
}

// (see kc_br_* above)  d_rows: n rows of EW words on the device.  *d_fix (scope-owned): nfix pairs {pid, link word}.  *tie: two different
// edges share a priority -- nothing was settled, the caller takes the host version.
static int settle_branches_device(kc_ctx *ctx, KcScope &scope, const u64 *d_rows, u64 n, int EW, u64 **d_fix, u64 *nfix, bool *tie, u64 *rounds_out) {
    *d_fix = nullptr; *nfix = 0; *tie = false;
    if (n >= ((u64)1 << 31)) { *tie = true; return KC_OK; }   // (edge ranks and port indices are 32 bits here)
    u64 *key = nullptr, *key2 = nullptr, *perm = nullptr, *perm2 = nullptr, *cnt = nullptr, *uedge = nullptr;
    u8 *keep = nullptr;
    KC_TRY(scope.alloc(&key, n + 1)); KC_TRY(scope.alloc(&key2, n + 1)); KC_TRY(scope.alloc(&perm, n + 1)); KC_TRY(scope.alloc(&perm2, n + 1));
    KC_TRY(scope.alloc(&cnt, 8)); KC_TRY(scope.alloc(&keep, n + 1)); KC_TRY(scope.alloc(&uedge, n + 1));
    KC_HIP(ctx, hipMemsetAsync(cnt, 0, 64, ctx->stream));
    KC_LAUNCH(kc_br_keys_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, d_rows, n, EW, key, perm);
    KC_TRY(kcu_sort_pairs(ctx, key, key2, perm, perm2, n, 0, 64, false));
    KC_LAUNCH(kc_br_mark_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, d_rows, (const u64 *)perm2, n, EW, keep, cnt);
    KC_TRY(kcu_select_flagged(ctx, perm2, keep, n, uedge, cnt + 1, false));
    u64 h[2] = {0, 0};
    KC_TRY(read_words(ctx, cnt, h, 2));
    if (h[0]) { *tie = true; return KC_OK; }
    const u64 ne = h[1];
    scope.release(key); scope.release(key2); scope.release(perm); scope.release(perm2); scope.release(keep);
    u64 *fix = nullptr;
    KC_TRY(scope.alloc(&fix, 4 * n + 2));   // (at most two entries per row)
    u64 *ports = nullptr, *ports2 = nullptr, *uports = nullptr, *akey = nullptr, *aval = nullptr, *akey2 = nullptr, *aval2 = nullptr;
    u8 *uf = nullptr, *isbranch = nullptr;
    u32 *mate = nullptr, *pick = nullptr, *emitted = nullptr;
    u64 np = 0;
    if (ne) {
        KC_TRY(scope.alloc(&ports, 2 * ne + 1)); KC_TRY(scope.alloc(&ports2, 2 * ne + 1)); KC_TRY(scope.alloc(&uports, 2 * ne + 1)); KC_TRY(scope.alloc(&uf, 2 * ne + 1));
        KC_LAUNCH(kc_br_ports_kernel, dim3(kc_grid(ne, 256)), dim3(256), 0, ctx->stream, d_rows, (const u64 *)uedge, ne, EW, ports);
        KC_TRY(kcu_sort_pairs(ctx, ports, ports2, nullptr, nullptr, 2 * ne, 0, 40, false));   // (port ids: 38 bits)
        KC_LAUNCH(kc_br_uniq_flags_kernel, dim3(kc_grid(2 * ne, 256)), dim3(256), 0, ctx->stream, (const u64 *)ports2, 2 * ne, uf);
        KC_TRY(kcu_select_flagged(ctx, ports2, uf, 2 * ne, uports, cnt + 2, false));
        KC_TRY(read_words(ctx, cnt + 2, &np, 1));
        scope.release(ports); scope.release(ports2); scope.release(uf);
        KC_TRY(scope.alloc(&akey, 2 * ne + 1)); KC_TRY(scope.alloc(&aval, 2 * ne + 1)); KC_TRY(scope.alloc(&akey2, 2 * ne + 1)); KC_TRY(scope.alloc(&aval2, 2 * ne + 1));
        KC_LAUNCH(kc_br_adj_kernel, dim3(kc_grid(ne, 256)), dim3(256), 0, ctx->stream, d_rows, (const u64 *)uedge, ne, EW, (const u64 *)uports, np, akey, aval);
        KC_TRY(kcu_sort_pairs(ctx, akey, akey2, aval, aval2, 2 * ne, 0, 64, false));
        scope.release(akey); scope.release(aval);
    }
    KC_TRY(scope.alloc(&mate, np + 1)); KC_TRY(scope.alloc(&pick, np + 1)); KC_TRY(scope.alloc(&emitted, np + 1)); KC_TRY(scope.alloc(&isbranch, np + 1));
    KC_HIP(ctx, hipMemsetAsync(mate, 0xff, (np + 1) * 4, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(emitted, 0, (np + 1) * 4, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(isbranch, 0, np + 1, ctx->stream));
    u64 rounds = 0;
    if (np) {
        KC_LAUNCH(kc_br_branchflag_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, d_rows, n, EW, (const u64 *)uports, np, isbranch);
        for (;; ++rounds) {
            KC_HIP(ctx, hipMemsetAsync(cnt + 3, 0, 8, ctx->stream));
            KC_LAUNCH(kc_br_pick_kernel, dim3(kc_grid(np, 256)), dim3(256), 0, ctx->stream, (const u64 *)akey2, (const u64 *)aval2, 2 * ne, np, (const u32 *)mate, pick);
            KC_LAUNCH(kc_br_accept_kernel, dim3(kc_grid(np, 256)), dim3(256), 0, ctx->stream, (const u64 *)akey2, (const u64 *)aval2, 2 * ne, np, (const u32 *)pick, mate, cnt + 3);
            u64 changed = 0;
            KC_TRY(read_words(ctx, cnt + 3, &changed, 1));
            if (!changed) break;
        }
    }
    KC_HIP(ctx, hipMemsetAsync(cnt + 4, 0, 8, ctx->stream));
    KC_LAUNCH(kc_br_emit_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, d_rows, n, EW, (const u64 *)uports, np, (const u32 *)mate, (const u8 *)isbranch, emitted, fix, cnt + 4);
    KC_HIP(ctx, hipGetLastError());
    KC_TRY(read_words(ctx, cnt + 4, nfix, 1));
    // (the stream is drained: everything but the result goes back to the pool now, not when the whole flow returns)
    scope.release(cnt); scope.release(uedge); scope.release(uports); scope.release(akey2); scope.release(aval2);
    scope.release(mate); scope.release(pick); scope.release(emitted); scope.release(isbranch);
    *d_fix = fix;
    if (rounds_out) *rounds_out = rounds;
    return KC_OK;
}

template <int W> int compute_sharded_w(kc_ctx *ctx, kc_comm *c, const u8 *d_seq, u64 len, int k, int complements, int want_mask, int min_frequency, kc_result **out) {
    KcScope scope(ctx);
    const int G = c->world, me = c->rank;
    const bool trace = trace_on();
    hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1], e2 = ctx->ev[2], e3 = ctx->ev[3];
    hipEvent_t eend;
    KC_HIP(ctx, hipEventCreate(&eend));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } evg{eend};
    KC_HIP(ctx, hipEventRecord(e0, ctx->stream));
    // KC_SHARD_TRACE: wall time of every step of the flow on rank 0 (stream drained at each mark), with the collectives and bytes so far
    auto t_mark = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        if (!trace || me != 0) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[shard 0/%d] step %-28s %8.3f ms   (collectives so far %llu, %.1f MB sent, %.3f ms inside them)\n", G, what,
                std::chrono::duration<double, std::milli>(now - t_mark).count(), (unsigned long long)c->calls, c->bytes_sent / 1e6, c->seconds * 1e3);
        t_mark = now;
    };
    // ---------------- phase 0
    kc_set *set = nullptr;
    const u64 occ_local = len >= (u64)k ? len - k + 1 : 0;
    KC_TRY(kcs_create_sharded(ctx, c, k, complements, occ_local, &set));
    struct SetGuard { kc_set *s; ~SetGuard() { if (s) kc_set_destroy(s); } } setg{set};
    {
        u64 *codes = nullptr, *brk = nullptr, *low = nullptr;
        int rc = kcs_pack(ctx, d_seq, len, &codes, &brk, &low);
        u64 moved = 0;
        if (rc == KC_OK) rc = kcs_build_sharded(set, c, codes, brk, low, len, &moved, min_frequency);
        (void)hipStreamSynchronize(ctx->stream);
        kc_release(ctx, codes); kc_release(ctx, brk); kc_release(ctx, low);
        KC_TRY(rc);
    }
    const u64 N_local = set->size_host;
    u64 N = N_local;
    KC_TRY(kcc_allreduce_sum(c, &N, 1));
    if (N == 0) return kc_fail_agreed(ctx, KC_E_NO_KMERS, "input contains no k-mers");
    if (c->test_fail_rank == me) return kc_fail(ctx, KC_E_NO_DEVICE, "test: this rank fails on its own");   // (kc_comm::test_fail_rank, a test knob: between two exchanges)
    KC_HIP(ctx, hipEventRecord(e1, ctx->stream));
    tick("set build (phase 0)");
    // Ports are numbered by DENSE node id (kc_set_dev.h): block * nstride + the k-mer's rank among the block's occupied slots, with
    // the same nstride on every rank -- so rank r holds the ids [r * rcap, (r + 1) * rcap) with rcap = blocks per rank * nstride, and
    // everything below addresses links, fragments and visit marks in that id space exactly as it used to in the slot space (at
    // half the bytes: 16 per k-mer instead of 16 per slot of a table at load 0.5).  rloS / rcapS are the table slots of this rank.
    KC_TRY(kcs_prepare_dense_sharded(set, c));
    SetView<W> sv = kc_view<W>(set);
    const u64 rloS = set->rlo, rcapS = set->rcap;
    const u32 nstride = set->nstride;
    const u64 rlo = (rloS >> set->bshift) * (u64)nstride, rcap = (rcapS >> set->bshift) * (u64)nstride;
    const u32 *ncount = sv.ncount;   // (biased: indexed by global block)
    // ---------------- phase 1a: links
    u64 *link_alloc = nullptr;
    KC_TRY(scope.alloc(&link_alloc, 2 * rcap));
    u64 *link = link_alloc - 2 * rlo;   // indexed with global port ids
    u64 *cur = nullptr;                  // cursors + flags
    KC_TRY(scope.alloc(&cur, (2 * KC_USH + 4) * KC_USTRIDE));
    ShLists L{};
    const int QW = W + 2, EW = 4 + 2 * W;
    // List capacities from the expected counts (+15 %): the home of a port's candidates leaves its source's block ~13 % of the
    // time and is then remote with probability (G-1)/G -- ~0.23 queries per k-mer at 8 ranks, 8(W+2) bytes each in the list.
    // An overflow doubles the lists and repeats the sweep, collectively.
    const double p_rem = 2.0 / (double)(k - set->mlen) * (double)(G - 1) / (double)G;   // (a middle holds k - m - 1 m-mers: its neighbour's minimizer differs with probability ~2 / (k - m))
    u64 qry_scap = (u64)(1.15 * 2.0 * p_rem * (double)N_local) / KC_USH + 4096;
    u64 edge_cap = N_local / 64 + ((u64)1 << 16), spill_cap = N_local / 64 + ((u64)1 << 16);
    u64 *spill = nullptr;
    constexpr int KSPEC = W == 1 ? 31 : (W == 2 ? 63 : 127);
    u64 nq = 0, n_branch_edges = 0;
    for (int attempt = 0;; ++attempt) {
        KC_TRY(scope.alloc(&L.qry, qry_scap * KC_USH * QW));
        KC_TRY(scope.alloc(&L.edges, edge_cap * EW + 1));
        KC_TRY(scope.alloc(&spill, spill_cap + 1));
        L.qry_scap = qry_scap; L.edge_cap = edge_cap;
        L.qry_cur = cur + KC_USH * KC_USTRIDE; L.flags = cur + 2 * KC_USH * KC_USTRIDE; L.edge_cur = cur + (2 * KC_USH + 1) * KC_USTRIDE;
        KC_HIP(ctx, hipMemsetAsync(cur, 0, (2 * KC_USH + 4) * KC_USTRIDE * 8, ctx->stream));
        KC_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
        {
            // the single-GPU round-1 kernel in its SH form: one workgroup per block of this rank (tags, and single-word keys, staged in LDS)
            const bool kspec = k == KSPEC && set->mlen == KC_MSPEC && !getenv("KC_NO_KSPEC");
            const bool stage_keys = W == 1 && set->bshift <= 13 && !getenv("KC_NOSTAGE");
            const int nt = stage_keys ? 1024 : 512;
            const bool mm = nt == 512;
            const size_t lds = kc_probe_lds_bytes(set->bshift, W, stage_keys, nt, mm);
            const u64 bpr = rcapS >> set->bshift;
            const unsigned pgrid = (unsigned)((bpr + 7) / 8 * 8);
            auto launch = [&](auto kern) -> int {
                KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                u64 *spill_cur = cur + (2 * KC_USH + 2) * KC_USTRIDE;
                KC_LAUNCH(kern, dim3(pgrid), dim3(nt), lds, ctx->stream, sv, link, k, complements, 1, (u64 *)nullptr, (u64)0, (u64 *)nullptr, (u64 *)nullptr, (u64)0,
                                   (u64 *)nullptr, spill, spill_cap, spill_cur, L);
                KC_LAUNCH((kc_shard_spill_kernel<W>), dim3(256), dim3(256), 0, ctx->stream, sv, link, k, complements, (const u64 *)spill, spill_cap, (const u64 *)spill_cur, L);
                return KC_OK;
            };
            // (GP: this rank's keys all sit at their primary homes -- the sweep guesses "in my own block" and computes no minimizers, see the
            // kernel; a rank decides that for itself, the results do not depend on it)
            const bool gp = set->homes_primary && !getenv("KC_PROBE_NO_GUESS");
            if (stage_keys) {
                if constexpr (W == 1) {
                    if (gp) KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, true, 1024, false, true, true>) : launch(kc_probe_block_kernel<W, 0, true, 1024, false, true, true>));
                    else KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, true, 1024, false, true>) : launch(kc_probe_block_kernel<W, 0, true, 1024, false, true>));
                }
            } else if (gp) KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, false, 512, true, true, true>) : launch(kc_probe_block_kernel<W, 0, false, 512, true, true, true>));
            else KC_TRY(kspec ? launch(kc_probe_block_kernel<W, KSPEC, false, 512, true, true>) : launch(kc_probe_block_kernel<W, 0, false, 512, true, true>));
        }
        KC_HIP(ctx, hipGetLastError());
        KC_HIP(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
        if (trace) { hipError_t e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[shard %d/%d] probe sweep done (%s), N_local=%llu\n", me, G, hipGetErrorString(e), (unsigned long long)N_local); }
        auto grow = [&]() { scope.release(L.qry); scope.release(L.edges); scope.release(spill); qry_scap = qry_scap * 3 + 4096; edge_cap = edge_cap * 4 + n_branch_edges; spill_cap *= 4; };
        {   // a list overflowed during the sweep: every rank repeats it with larger lists (nothing has been exchanged yet)
            u64 any = 0;
            KC_TRY(read_words(ctx, L.flags, &any, 1));
            KC_TRY(kcc_allreduce_max(c, &any, 1));
            if (any) {
                if (attempt == 4) return kc_fail(ctx, KC_E_CAPACITY, "hash-sharded probe: query / branch edge lists overflow");
                KC_TRY(read_words(ctx, L.edge_cur, &n_branch_edges, 1));
                grow();
                continue;
            }
        }
        // queries -> owners -> link words back
        nq = 0;
        {
            u64 *qc = nullptr;
            bool ovf = false;
            std::vector<u64> hc(KC_USH * KC_USTRIDE);
            KC_TRY(read_words(ctx, L.qry_cur, hc.data(), KC_USH * KC_USTRIDE));
            for (int sh = 0; sh < KC_USH; ++sh) nq += hc[(size_t)sh * KC_USTRIDE] > qry_scap ? qry_scap : hc[(size_t)sh * KC_USTRIDE];
            KC_TRY(scope.alloc(&qc, nq * QW + 1));
            KC_TRY(kcu_compact_shards(ctx, L.qry, qry_scap, L.qry_cur, KC_USH, KC_USTRIDE, QW, qc, &nq, &ovf));
            u8 *dest = nullptr;
            KC_TRY(scope.alloc(&dest, nq + 1));
            if (nq) KC_LAUNCH((kc_shard_qdest_kernel<W>), dim3(kc_grid(nq, 256)), dim3(256), 0, ctx->stream, (const u64 *)qc, nq, rcapS, dest);
            KcRoute rt;
            KC_TRY(kcr_build(c, dest, nq, &rt));
            u64 *rq = nullptr, *ans = nullptr, *ansb = nullptr;
            int rc = kcr_forward(c, rt, qc, W + 1, QW, &rq);   // the k-mer and the asking port travel; the owner derives the home (one minimizer pass)
            if (rc == KC_OK) rc = kc_alloc(ctx, &ans, rt.n_recv + 1);
            if (rc == KC_OK && rt.n_recv)
                KC_LAUNCH((kc_shard_answer_kernel<W>), dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, sv, k, complements, (const u64 *)rq, rt.n_recv, ans, L);
            if (trace) { hipError_t e = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[shard %d/%d] %llu queries out, %llu answered here (%s)\n", me, G, (unsigned long long)nq, (unsigned long long)rt.n_recv, hipGetErrorString(e)); }
            if (rc == KC_OK) rc = kc_alloc(ctx, &ansb, nq + 1);
            if (rc == KC_OK) rc = kcr_backward(c, rt, ans, 1, ansb);
            if (rc == KC_OK && nq)
                KC_LAUNCH(kc_shard_fill_kernel, dim3(kc_grid(nq, 256)), dim3(256), 0, ctx->stream, (const u64 *)qc, (const u64 *)ansb, nq, QW, link, nstride, set->bshift);
            (void)hipStreamSynchronize(ctx->stream);
            kcr_free(ctx, &rt);
            kc_release(ctx, rq); kc_release(ctx, ans); kc_release(ctx, ansb);
            scope.release(qc); scope.release(dest);
            KC_TRY(rc);
        }
        u64 ovf = 0;
        KC_TRY(read_words(ctx, L.flags, &ovf, 1));
        KC_TRY(read_words(ctx, L.edge_cur, &n_branch_edges, 1));
        u64 any = ovf;
        KC_TRY(kcc_allreduce_max(c, &any, 1));   // (the retry is collective: every rank repeats sweep and exchange)
        if (!any) break;
        if (attempt == 4) return kc_fail(ctx, KC_E_CAPACITY, "hash-sharded probe: query / branch edge lists overflow");
        grow();   // (the owners' branch edge lists overflowed while answering: sweep and exchange once more)
    }
    scope.release(L.qry); L.qry = nullptr;
    scope.release(spill);
    // the branch subgraph: every rank's edges to every rank, settled identically everywhere
    {
        u64 *all = nullptr, n_all = 0;
        KC_TRY(allgather_rows(c, L.edges, n_branch_edges, EW, &all, &n_all, nullptr));
        scope.release(L.edges); L.edges = nullptr;
        bool on_host = getenv("KC_BRANCH_HOST") != nullptr;   // (test knob: the host version)
        if (n_all && !on_host) {
            u64 *dfix = nullptr, nfix = 0, brounds = 0;
            KC_TRY(settle_branches_device(ctx, scope, all, n_all, EW, &dfix, &nfix, &on_host, &brounds));
            if (!on_host) {
                if (nfix) KC_LAUNCH((kc_shard_apply_fix_kernel<W>), dim3(kc_grid(nfix, 256)), dim3(256), 0, ctx->stream, sv, link, (const u64 *)dfix, nfix, 2 * rlo, 2 * rcap);
                KC_HIP(ctx, hipGetLastError());
                KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                scope.release(dfix);
                if (trace && me == 0) fprintf(stderr, "[shard 0/%d] branch subgraph on the device: %llu rows, %llu link words change, %llu matching rounds\n", G, (unsigned long long)n_all,
                                              (unsigned long long)nfix, (unsigned long long)brounds);
                n_branch_edges = n_all;
            }
        }
        if (n_all && on_host) {
            std::vector<u64> rows(n_all * EW), fix;
            KC_HIP(ctx, hipMemcpyAsync(rows.data(), all, rows.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            settle_branches<W>(rows, n_all, fix);
            if (!fix.empty()) {
                u64 *dfix = nullptr;
                KC_TRY(scope.alloc(&dfix, fix.size()));
                KC_HIP(ctx, hipMemcpyAsync(dfix, fix.data(), fix.size() * 8, hipMemcpyHostToDevice, ctx->stream));
                KC_LAUNCH((kc_shard_apply_fix_kernel<W>), dim3(kc_grid(fix.size() / 2, 256)), dim3(256), 0, ctx->stream, sv, link, (const u64 *)dfix, (u64)(fix.size() / 2),
                                   2 * rlo, 2 * rcap);
                KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                scope.release(dfix);
            }
            n_branch_edges = n_all;
        }
        kc_release(ctx, all);
    }
    if (trace) fprintf(stderr, "[shard %d/%d] N_local=%llu N=%llu queries=%llu branch edges (all ranks)=%llu\n", me, G, (unsigned long long)N_local,
                       (unsigned long long)N, (unsigned long long)nq, (unsigned long long)n_branch_edges);
    tick("links (probe, queries, branch)");
    // ---------------- phase 1b: fragments (the discovery restarts after a cycle has been cut)
    u64 *entries = nullptr, n_ent = 0;
    u64 *f_exit = nullptr, *f_partner = nullptr, *f_len = nullptr, *nxt = nullptr, *dist = nullptr, *last = nullptr, *iota = nullptr, *act = nullptr;
    u32 *f_stream = nullptr, *fidx = nullptr, *chunk_next = nullptr;
    u64 *pool = nullptr;
    u8 *flags8 = nullptr;
    KC_TRY(scope.alloc(&fidx, 2 * rcap));
    u64 rounds = 0, cycles_cut = 0;
    // in-block runs contracted before the fragment walks (kc_frag_contract_kernel over this rank's blocks; the replaced link words are
    // logged): as soon as anything has to walk or cut plain links (cycles) the words are put back and the plain walk takes over
    bool frag_ok = set->mlen && set->bshift <= 13 && !getenv("KC_NOFRAG"), contracted = false;
    u64 *fpool = nullptr, *fbase = nullptr, *fcur = nullptr, *flog = nullptr, *fblog = nullptr;
    const u64 bpr_blocks = rcapS >> set->bshift;
    auto restore_links = [&]() -> int {
        if (!contracted) return KC_OK;
        KC_LAUNCH(kc_frag_restore_kernel, dim3((unsigned)bpr_blocks), dim3(256), 0, ctx->stream, link, (const u64 *)flog, (const u64 *)fblog, set->bshift, rloS >> set->bshift, nstride);
        KC_HIP(ctx, hipGetLastError());
        contracted = false; frag_ok = false;
        return KC_OK;
    };
    for (int pass = 0;; ++pass) {
        if (pass == 64) return kc_fail(ctx, KC_E_NO_DEVICE, "chain discovery does not settle after cutting cycles");
        // entries, fragment walks.  K-mers that no fragment reaches lie on cycles inside this rank: counted here, cut after the ranking
        u64 missing = 0, walk_err = 0;
        {
            // entries: listed by the contraction kernel on its way (it has every link word of the block in registers), or by a sweep of
            // their own (no contraction: cycles, KC_NOFRAG; or more entries than the list was sized for)
            KcFrag fr{nullptr, nullptr, 0, 0};
            bool have_entries = false;
            if (frag_ok && !contracted && bpr_blocks) {
                u64 pool_cap = (rcapS / 16 + 65536) / 64 * 64;
                if (pool_cap > ((u64)1 << 29)) pool_cap = (u64)1 << 29;   // (a shortcut names a pool run by its absolute first word in 29 bits)
                const u64 log_cap = (rcapS / 6 + 65536) / 64 * 64, ecap = N_local / 2 + 65536;   // (per-block shares, as on one GPU)
                KC_TRY(scope.alloc(&fpool, pool_cap)); KC_TRY(scope.alloc(&fbase, bpr_blocks)); KC_TRY(scope.alloc(&fcur, 2 * 64 * 16 + 16));
                KC_TRY(scope.alloc(&flog, log_cap)); KC_TRY(scope.alloc(&fblog, bpr_blocks));
                KC_TRY(scope.alloc(&entries, ecap + 1));
                KC_HIP(ctx, hipMemsetAsync(fcur, 0, (2 * 64 * 16 + 16) * 8, ctx->stream));
                const size_t flds = (size_t)2 * ((size_t)1 << set->bshift) * 4 + KC_FRAG_LONGS * 8 + KC_FRAG_ENTRIES * 2;
                KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_frag_contract_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
                KC_LAUNCH((kc_frag_contract_kernel<W, true>), dim3((unsigned)bpr_blocks), dim3(1024), flds, ctx->stream, sv, link, complements, fpool, pool_cap, fcur, fbase, flog,
                                   log_cap, fcur + 64 * 16, fblog, entries, ecap, fcur + 2 * 64 * 16);
                KC_HIP(ctx, hipGetLastError());
                contracted = true;
                fr = KcFrag{fpool, fbase, 0, rloS >> set->bshift};
                KC_TRY(read_words(ctx, fcur + 2 * 64 * 16, &n_ent, 1));
                have_entries = n_ent <= ecap;
                if (!have_entries) scope.release(entries);
            } else if (contracted) fr = KcFrag{fpool, fbase, 0, rloS >> set->bshift};
            if (!have_entries) {
                const u64 ntiles = (rcap + 2047) / 2048;
                u64 *tcount = nullptr;
                unsigned short *emasks = nullptr;
                KC_TRY(scope.alloc(&tcount, ntiles + 2));
                KC_TRY(scope.alloc(&emasks, ntiles * 256));
                KC_LAUNCH((kc_shard_entries_kernel<false>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, (const u64 *)link, ncount, nstride, rlo, rcap, complements,
                                   (const u64 *)nullptr, tcount, emasks);
                KC_TRY(kcu_exclusive_scan(ctx, tcount, tcount, ntiles));
                KC_TRY(read_words(ctx, tcount + ntiles, &n_ent, 1));
                if (n_ent >= ((u64)1 << 32)) return kc_fail(ctx, KC_E_CAPACITY, "more than 2^32 chain fragments on one rank");
                KC_TRY(scope.alloc(&entries, n_ent + 1));
                KC_LAUNCH((kc_shard_entries_kernel<true>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, (const u64 *)link, ncount, nstride, rlo, rcap, complements,
                                   (const u64 *)tcount, entries, emasks);
                KC_HIP(ctx, hipGetLastError());
                KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                scope.release(tcount); scope.release(emasks);
            }
            if (n_ent >= ((u64)1 << 32)) return kc_fail(ctx, KC_E_CAPACITY, "more than 2^32 chain fragments on one rank");
            KC_TRY(scope.alloc(&f_exit, n_ent + 1)); KC_TRY(scope.alloc(&f_partner, n_ent + 1)); KC_TRY(scope.alloc(&f_len, n_ent + 1));
            KC_TRY(scope.alloc(&f_stream, n_ent + 1));
            FragRec frec{};
            frec.pool_cap = n_ent + 2 * N_local / (32 * KC_FCW) + N_local / 8 + 1024;   // (first chunks by walk index, later ones in batches of 8 of which up to 7 stay unused)
            KC_TRY(scope.alloc(&pool, frec.pool_cap * KC_FCW)); KC_TRY(scope.alloc(&chunk_next, frec.pool_cap));
            frec.pool = pool; frec.chunk_next = chunk_next; frec.pool_cursor = cur;
            KC_HIP(ctx, hipMemsetAsync(cur, 0, 4 * KC_USTRIDE * 8, ctx->stream));
            if (n_ent) {
                if (contracted)
                    KC_LAUNCH((kc_shard_walk_kernel<false, true>), dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, rlo, rcap, (const u64 *)entries, n_ent,
                                       f_exit, f_partner, f_len, f_stream, fidx, frec, (u8 *)nullptr, cur + KC_USTRIDE, fr, set->bshift);
                else
                    KC_LAUNCH((kc_shard_walk_kernel<false, false>), dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, rlo, rcap, (const u64 *)entries, n_ent,
                                       f_exit, f_partner, f_len, f_stream, fidx, frec, (u8 *)nullptr, cur + KC_USTRIDE, fr, set->bshift);
                KC_LAUNCH(kc_shard_sum_kernel, dim3(256), dim3(256), 0, ctx->stream, (const u64 *)f_len, n_ent, cur + 2 * KC_USTRIDE);
            }
            KC_HIP(ctx, hipGetLastError());
            u64 hw[3 * KC_USTRIDE];
            KC_TRY(read_words(ctx, cur, hw, 3 * KC_USTRIDE));
            if (hw[KC_USTRIDE]) walk_err = 1;
            const u64 walked = hw[2 * KC_USTRIDE] / (complements ? 2 : 1);
            if (walked > N_local) walk_err = 2;
            missing = walk_err ? 0 : N_local - walked;
        }
        {   // (a rank that returned alone here would leave its peers waiting in the ranking's exchanges: the ranks agree first)
            u64 e = walk_err;
            KC_TRY(kcc_allreduce_max(c, &e, 1));
            if (e == 1) return kc_fail_agreed(ctx, KC_E_CAPACITY, "fragment recording pool exhausted (on some rank)");
            if (e) return kc_fail_agreed(ctx, KC_E_NO_DEVICE, "fragment walks and k-mer count disagree (on some rank)");
        }
        KC_TRY(scope.alloc(&nxt, n_ent + 1)); KC_TRY(scope.alloc(&dist, n_ent + 1)); KC_TRY(scope.alloc(&last, n_ent + 1));
        KC_TRY(scope.alloc(&iota, n_ent + 1)); KC_TRY(scope.alloc(&act, n_ent + 1)); KC_TRY(scope.alloc(&flags8, n_ent + 1));
        if (n_ent) {
            KC_LAUNCH(kc_shard_jump_init_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)f_partner, (const u64 *)f_len, n_ent, me, nxt, dist, last);
            KC_LAUNCH(kc_shard_iota_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, iota, n_ent);
        }
        KC_HIP(ctx, hipGetLastError());
        tick("contraction + fragment walks");
        // ---- ranking by waves (see kc_shard_wave_init_kernel) when the set is bidirectional and the chains are short in fragments
        u64 prev_active = ~(u64)0, n_act = 0;
        bool stalled = false, ranked = false;
        {
            u64 tot[2] = {n_ent, 0};   // fragments, chain heads (counted below) over all ranks
            u64 wave_max = 2048;
            if (const char *e = getenv("KC_WAVE_MAX")) wave_max = (u64)atoll(e);   // 0: always pointer jumping (tests)
            u64 *pay = nullptr, *poff = nullptr, *msgs = nullptr;
            u8 *reach_a = nullptr, *reach_b = nullptr, *wdest = nullptr;
            KC_TRY(scope.alloc(&pay, n_ent + 1)); KC_TRY(scope.alloc(&poff, n_ent + 1)); KC_TRY(scope.alloc(&reach_a, n_ent + 1)); KC_TRY(scope.alloc(&reach_b, n_ent + 1));
            if (n_ent) KC_LAUNCH(kc_shard_wave_init_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, (const u64 *)entries, n_ent, (const u64 *)f_exit,
                                          (const u32 *)fidx, 2 * rlo, me, pay, poff, reach_a);
            if (n_ent) {
                KC_TRY(kcu_select_flagged(ctx, iota, reach_a, n_ent, act, cur + 2 * KC_USTRIDE));
                KC_TRY(read_words(ctx, cur + 2 * KC_USTRIDE, &tot[1], 1));
            }
            KC_TRY(kcc_allreduce_sum(c, tot, 2));
            // a chain has tot[0] / tot[1] fragments on average: waves only when that many rounds are affordable
            bool wave = complements && wave_max && tot[1] && tot[0] / tot[1] <= wave_max / 8;
            u64 wrounds = 0;
            // ---- sampled ranking (see kc_shard_sr_init_kernel): when the plain waves would need more rounds than R + log2 of it.
            // KC_RANK_SAMPLED=0 / 1: never / always (tests); KC_RANK_C, KC_RANK_R: splitter rate and wave rounds.
            u32 SR_C = 8, SR_R = 24;
            if (const char *e = getenv("KC_RANK_C")) { const int v = atoi(e); if (v >= 1 && v <= 4096) SR_C = (u32)v; }
            if (const char *e = getenv("KC_RANK_R")) { const int v = atoi(e); if (v >= 1 && v <= 4096) SR_R = (u32)v; }
            bool sampled = complements && wave_max && tot[1] && tot[0] / tot[1] > 2 * SR_R / 3;
            if (const char *e = getenv("KC_RANK_SAMPLED")) sampled = complements && wave_max && atoi(e) != 0;
            if (sampled) {
                u64 *sjump = nullptr, *act2 = nullptr, *items = nullptr, *slotpos = nullptr, *ansbuf = nullptr, *ansback = nullptr;
                u8 *spl = nullptr, *sdest = nullptr;
                KC_TRY(scope.alloc(&sjump, n_ent + 1)); KC_TRY(scope.alloc(&act2, n_ent + 1)); KC_TRY(scope.alloc(&items, 2 * (n_ent + 1)));
                KC_TRY(scope.alloc(&slotpos, n_ent + 1)); KC_TRY(scope.alloc(&spl, n_ent + 1)); KC_TRY(scope.alloc(&sdest, n_ent + 1));
                u64 *acur = cur + 3 * KC_USTRIDE;   // [0], [1]: cursors of the two active lists, [2]: a wave message did not fit its two words
                KC_HIP(ctx, hipMemsetAsync(acur, 0, 24, ctx->stream));
                if (n_ent) KC_LAUNCH(kc_shard_sr_init_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, (const u64 *)entries, n_ent, (const u64 *)f_exit,
                                              (const u64 *)f_partner, (const u32 *)fidx, 2 * rlo, me, SR_C, pay, poff, sjump, spl, act, acur);
                u64 n_cur = 0;
                KC_TRY(read_words(ctx, acur, &n_cur, 1));
                // slot sizes come from replicated values only: the largest rank's counts now, the previous exchange's maximum later
                u64 mx[2] = {n_cur, n_ent};
                KC_TRY(kcc_allreduce_max(c, mx, 2));
                const u64 peers = G > 1 ? (u64)(G - 1) : 1;
                const u64 cap0 = mx[0] / peers + mx[0] / (4 * peers) + 1024;
                SlotX sx(ctx, c, &scope);
                u64 *la = act, *lb = act2;   // this round's senders, the next round's
                int ca = 0;                  // cursor of la
                u64 h4[4] = {0, 0, 0, 0};
                u64 capw = cap0;
                for (u32 r = 1; r <= SR_R + 1; ++r) {
                    if (r == SR_R + 1) {
                        if (n_ent) KC_LAUNCH(kc_shard_sr_unreached_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, n_ent, (const u64 *)f_partner, (const u64 *)pay, spl, la, acur + ca);
                        KC_TRY(read_words(ctx, acur + ca, &n_cur, 1));
                        capw = cap0;
                    }
                    KC_TRY(sx.ensure(2, capw));
                    if (n_cur) KC_LAUNCH(kc_shard_sr_send_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)la, n_cur, (const u64 *)f_partner, (const u64 *)f_len,
                                                  (const u64 *)pay, (const u64 *)poff, (const u8 *)spl, me, 2 * rcap, items, sdest, acur + 2);
                    KC_TRY(sx.exchange(items, sdest, n_cur, nullptr, h4));
                    ++rounds;
                    KC_HIP(ctx, hipMemsetAsync(acur + (ca ^ 1), 0, 8, ctx->stream));
                    if (h4[3]) KC_LAUNCH(kc_shard_sr_recv_kernel, dim3(kc_grid((u64)G * sx.cap, 256)), dim3(256), 0, ctx->stream, (const u64 *)sx.recv, sx.stride, sx.cap, G, (const u32 *)fidx,
                                                  2 * rlo, (const u64 *)f_partner, (const u8 *)spl, pay, poff, sjump, lb, acur + (ca ^ 1));
                    KC_TRY(read_words(ctx, acur + (ca ^ 1), &n_cur, 1));
                    std::swap(la, lb); ca ^= 1;
                    capw = h4[1] + h4[1] / 16 + 256;   // (waves only end: a round rarely sends more to a peer than the one before -- if one does, the exchange repeats)
                    if (h4[2] == 0 && r < SR_R) r = SR_R;   // every wave has ended early (all ranks see the same total): straight to the last round
                }
                {   // (the ranks find out together)
                    u64 bad = 0;
                    KC_TRY(read_words(ctx, acur + 2, &bad, 1));
                    KC_TRY(kcc_allreduce_max(c, &bad, 1));
                    if (bad) return kc_fail_agreed(ctx, KC_E_CAPACITY, "sampled fragment ranking: a wave message does not fit its two words (on some rank)");
                }
                // the reduced list: pointer jumping towards the heads
                KC_HIP(ctx, hipMemsetAsync(acur, 0, 16, ctx->stream));
                la = act; lb = act2; ca = 0;
                if (n_ent) KC_LAUNCH(kc_shard_sr_list_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, n_ent, (const u8 *)spl, (const u64 *)sjump, (const u64 *)pay, 0, la, acur);
                KC_TRY(read_words(ctx, acur, &n_cur, 1));
                u64 tspl[2] = {n_cur, n_cur};
                KC_TRY(kcc_allreduce_sum(c, tspl, 1));
                KC_TRY(kcc_allreduce_max(c, tspl + 1, 1));
                int max_jumps = 2;
                for (u64 v = tspl[0]; v > 1; v >>= 1) ++max_jumps;   // (a chain has at most that many splitters: log2 of it rounds reach any head)
                capw = tspl[1] / peers + tspl[1] / (4 * peers) + 1024;
                for (int j = 0; tspl[0] != 0; ++j) {
                    KC_TRY(sx.ensure(1, capw));
                    if (n_cur) KC_LAUNCH(kc_shard_sr_req_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)la, n_cur, (const u64 *)sjump, items, sdest);
                    KC_TRY(sx.exchange(items, sdest, n_cur, slotpos, h4));
                    ++rounds;
                    if (h4[2] == 0) break;   // nobody asks any more
                    const u64 aw = (u64)G * sx.cap * 2 + 1;
                    KC_TRY(scope.alloc(&ansbuf, aw)); KC_TRY(scope.alloc(&ansback, aw));
                    if (h4[3]) KC_LAUNCH(kc_shard_sr_ans_kernel, dim3(kc_grid((u64)G * sx.cap, 256)), dim3(256), 0, ctx->stream, (const u64 *)sx.recv, sx.stride, sx.cap, G, (const u64 *)sjump,
                                                  (const u64 *)poff, (const u64 *)pay, ansbuf);
                    KC_TRY(sx.answer(ansbuf, 2, ansback));
                    KC_HIP(ctx, hipMemsetAsync(acur + (ca ^ 1), 0, 8, ctx->stream));
                    if (j + 1 >= max_jumps) {   // still jumping after log2(all splitters) rounds: on a cycle
                        if (n_cur) KC_LAUNCH(kc_shard_sr_giveup_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)la, n_cur, pay);
                        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                        scope.release(ansbuf); scope.release(ansback);
                        break;
                    }
                    if (n_cur) KC_LAUNCH(kc_shard_sr_apply_jump_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)la, n_cur, (const u64 *)slotpos, (const u64 *)ansback,
                                                  sjump, poff, pay, lb, acur + (ca ^ 1));
                    KC_TRY(read_words(ctx, acur + (ca ^ 1), &n_cur, 1));
                    scope.release(ansbuf); scope.release(ansback);
                    std::swap(la, lb); ca ^= 1;
                    capw = h4[1] + h4[1] / 8 + 256;
                }
                // every other fragment pulls payload and nodes-before of its splitter
                KC_HIP(ctx, hipMemsetAsync(acur, 0, 16, ctx->stream));
                if (n_ent) KC_LAUNCH(kc_shard_sr_list_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, n_ent, (const u8 *)spl, (const u64 *)sjump, (const u64 *)pay, 1, act, acur);
                KC_TRY(read_words(ctx, acur, &n_cur, 1));
                {
                    KC_TRY(sx.ensure(1, mx[1] / peers + mx[1] / (4 * peers) + 1024));
                    if (n_cur) KC_LAUNCH(kc_shard_sr_req_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_cur, (const u64 *)pay, items, sdest);
                    KC_TRY(sx.exchange(items, sdest, n_cur, slotpos, h4));
                    const u64 aw = (u64)G * sx.cap * 2 + 1;
                    KC_TRY(scope.alloc(&ansbuf, aw)); KC_TRY(scope.alloc(&ansback, aw));
                    if (h4[3]) KC_LAUNCH(kc_shard_sr_ans_kernel, dim3(kc_grid((u64)G * sx.cap, 256)), dim3(256), 0, ctx->stream, (const u64 *)sx.recv, sx.stride, sx.cap, G, (const u64 *)sjump,
                                                  (const u64 *)poff, (const u64 *)pay, ansbuf);
                    KC_TRY(sx.answer(ansbuf, 2, ansback));
                    if (n_cur) KC_LAUNCH(kc_shard_sr_apply_pull_kernel, dim3(kc_grid(n_cur, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_cur, (const u64 *)slotpos, (const u64 *)ansback, poff, pay);
                    KC_HIP(ctx, hipGetLastError());
                    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    scope.release(ansbuf); scope.release(ansback);
                    rounds += 1;
                }
                if (trace && me == 0) fprintf(stderr, "[shard 0/%d] sampled ranking: C=%u R=%u, %llu slotted exchanges (%llu repeated after an overflow)\n", G, SR_C, SR_R,
                                              (unsigned long long)sx.exchanges, (unsigned long long)sx.redone);
                scope.release(sx.send); scope.release(sx.recv); scope.release(sx.cnt);
                scope.release(sjump); scope.release(act2); scope.release(items); scope.release(slotpos); scope.release(spl); scope.release(sdest);
                wave = true;   // (what follows the plain waves -- kc_shard_wave_finish_kernel -- follows here too)
            }
            while (wave && !sampled) {
                u64 n_send = 0;
                if (n_ent) {
                    KC_LAUNCH(kc_shard_wave_flag_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u8 *)reach_a, (const u64 *)f_partner, n_ent, flags8);
                    KC_TRY(kcu_select_flagged(ctx, iota, flags8, n_ent, act, cur + 2 * KC_USTRIDE));
                    KC_TRY(read_words(ctx, cur + 2 * KC_USTRIDE, &n_send, 1));
                }
                KC_TRY(scope.alloc(&msgs, 3 * n_send + 1)); KC_TRY(scope.alloc(&wdest, n_send + 1));
                if (n_send) KC_LAUNCH(kc_shard_wave_send_kernel, dim3(kc_grid(n_send, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_send, (const u64 *)f_partner,
                                               (const u64 *)f_len, (const u64 *)pay, (const u64 *)poff, 2 * rcap, msgs, wdest);
                KcRoute rt;
                KC_TRY(kcr_build(c, wdest, n_send, &rt));   // (its count exchange also says whether any rank still sends: no collective of its own for that)
                if (rt.n_total == 0 || ++wrounds > wave_max) {
                    if (rt.n_total) wave = false;   // a chain much longer than the average: pointer jumping after all
                    kcr_free(ctx, &rt);
                    scope.release(msgs); scope.release(wdest);
                    break;
                }
                u64 *rq = nullptr;
                int rc = kcr_forward(c, rt, msgs, 3, 3, &rq);
                if (rc == KC_OK && n_ent && hipMemsetAsync(reach_b, 0, n_ent, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
                if (rc == KC_OK && rt.n_recv)
                    KC_LAUNCH(kc_shard_wave_recv_kernel, dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, (const u64 *)rq, rt.n_recv, (const u32 *)fidx, 2 * rlo, pay, poff, reach_b);
                (void)hipStreamSynchronize(ctx->stream);
                kcr_free(ctx, &rt);
                kc_release(ctx, rq);
                scope.release(msgs); scope.release(wdest);
                KC_TRY(rc);
                std::swap(reach_a, reach_b);
                ++rounds;
            }
            if (wave) {
                // every chain with ends is ranked; fragments no wave reached lie on cycles that span ranks
                if (n_ent) {
                    KC_LAUNCH(kc_shard_wave_finish_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, n_ent, (const u64 *)f_exit, (const u32 *)fidx, 2 * rlo,
                                       (const u64 *)f_len, (const u64 *)f_partner, (const u64 *)pay, (const u64 *)poff, dist, last, nxt, flags8);
                    KC_TRY(kcu_select_flagged(ctx, iota, flags8, n_ent, act, cur + 2 * KC_USTRIDE));
                    KC_TRY(read_words(ctx, cur + 2 * KC_USTRIDE, &n_act, 1));
                }
                u64 tot_un = n_act;
                KC_TRY(kcc_allreduce_sum(c, &tot_un, 1));
                ranked = true;
                stalled = tot_un != 0;
            }
            scope.release(pay); scope.release(poff); scope.release(reach_a); scope.release(reach_b);
        }
        // pointer jumping (unidirectional sets, long chains)
        for (int round = 0; !ranked; ++round) {
            n_act = 0;
            if (n_ent) {
                KC_LAUNCH(kc_shard_active_flags_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)nxt, n_ent, flags8);
                KC_TRY(kcu_select_flagged(ctx, iota, flags8, n_ent, act, cur + 2 * KC_USTRIDE));
                KC_TRY(read_words(ctx, cur + 2 * KC_USTRIDE, &n_act, 1));
            }
            u64 tot_act = n_act;
            KC_TRY(kcc_allreduce_sum(c, &tot_act, 1));
            if (tot_act == 0) break;
            if (round > 1 && tot_act >= prev_active) { stalled = true; break; }   // only fragments on cycles are left
            if (round > 0) prev_active = tot_act;
            u64 *req = nullptr, *rq = nullptr, *ans = nullptr, *ansb = nullptr;
            u8 *dest = nullptr;
            const int A = round == 0 ? 1 : 2;
            KC_TRY(scope.alloc(&req, n_act + 1)); KC_TRY(scope.alloc(&dest, n_act + 1));
            if (n_act) KC_LAUNCH(kc_shard_req_kernel, dim3(kc_grid(n_act, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_act, (const u64 *)nxt, round == 0 ? 1 : 0, 2 * rcap, req, dest);
            KcRoute rt;
            KC_TRY(kcr_build(c, dest, n_act, &rt));
            int rc = kcr_forward(c, rt, req, 1, 1, &rq);
            if (rc == KC_OK) rc = kc_alloc(ctx, &ans, rt.n_recv * A + 1);
            if (rc == KC_OK && rt.n_recv) {
                if (round == 0) KC_LAUNCH(kc_shard_ans_port_kernel, dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, (const u64 *)rq, rt.n_recv, (const u32 *)fidx, 2 * rlo, me, ans);
                else KC_LAUNCH(kc_shard_ans_jump_kernel, dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, (const u64 *)rq, rt.n_recv, (const u64 *)nxt, (const u64 *)dist, (const u64 *)last, ans);
            }
            if (rc == KC_OK) rc = kc_alloc(ctx, &ansb, n_act * A + 1);
            if (rc == KC_OK) rc = kcr_backward(c, rt, ans, A, ansb);
            if (rc == KC_OK && n_act) {
                if (round == 0) KC_LAUNCH(kc_shard_apply_port_kernel, dim3(kc_grid(n_act, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_act, (const u64 *)ansb, nxt);
                else KC_LAUNCH(kc_shard_apply_jump_kernel, dim3(kc_grid(n_act, 256)), dim3(256), 0, ctx->stream, (const u64 *)act, n_act, (const u64 *)ansb, nxt, dist, last);
            }
            (void)hipStreamSynchronize(ctx->stream);
            kcr_free(ctx, &rt);
            kc_release(ctx, rq); kc_release(ctx, ans); kc_release(ctx, ansb);
            scope.release(req); scope.release(dest);
            KC_TRY(rc);
            ++rounds;
        }
        tick("fragment ranking");
        {
            u64 any_missing = missing;
            KC_TRY(kcc_allreduce_max(c, &any_missing, 1));
            if (!stalled && !any_missing) break;
        }
        // cycles (inside ranks and across them): every rank lists its k-mers that are on no chain with ends, all ranks take the same
        // decisions (kc_cycles_decide), the owners open the cycles, and the discovery starts over
        KC_TRY(restore_links());   // (everything below walks plain links, the cut rewrites them)
        {
            const int RW = 3 + W;
            u8 *visited = nullptr;
            u64 *ulist = nullptr, *rows = nullptr, *all = nullptr, *cand = nullptr, *allc = nullptr, n_all = 0, n_allc = 0, nunvis = 0, ncand = 0;
            KC_TRY(scope.alloc(&visited, rcap + 1));
            KC_HIP(ctx, hipMemsetAsync(visited, 0, rcap, ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(cur, 0, 4 * KC_USTRIDE * 8, ctx->stream));
            if (n_ent) KC_LAUNCH(kc_shard_mark_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, rlo, rcap, (const u64 *)entries, n_ent, (const u8 *)flags8, visited);
            KC_LAUNCH((kc_shard_unvisited_kernel<false>), kc_grid2(rcap, 256), dim3(256), 0, ctx->stream, ncount, nstride, (const u8 *)visited, rlo, rcap, (u64 *)nullptr, cur + 3 * KC_USTRIDE);
            KC_HIP(ctx, hipGetLastError());
            KC_TRY(read_words(ctx, cur + 3 * KC_USTRIDE, &nunvis, 1));
            KC_TRY(scope.alloc(&ulist, nunvis + 1)); KC_TRY(scope.alloc(&rows, (nunvis + 1) * RW));
            KC_HIP(ctx, hipMemsetAsync(cur + 3 * KC_USTRIDE, 0, 8, ctx->stream));
            if (nunvis) {
                KC_LAUNCH((kc_shard_unvisited_kernel<true>), kc_grid2(rcap, 256), dim3(256), 0, ctx->stream, ncount, nstride, (const u8 *)visited, rlo, rcap, ulist, cur + 3 * KC_USTRIDE);
                KC_LAUNCH((kc_cycle_rows_kernel<W, true>), kc_grid2(nunvis, 256), dim3(256), 0, ctx->stream, sv, (const u64 *)link, (const u64 *)ulist, nunvis, rows);
            }
            KC_HIP(ctx, hipGetLastError());
            {
                // every rank holds ALL rows (and walks them on the host): bounded, and the ranks find out together.  (2^27 k-mers on
                // cycles are ~5 GB of rows per rank; an input like that is not what the cycle rule was made for.)
                u64 tot = nunvis;
                KC_TRY(kcc_allreduce_sum(c, &tot, 1));
                if (tot > ((u64)1 << 27)) return kc_fail_agreed(ctx, KC_E_CAPACITY, "too many k-mers on cycles for the replicated cycle rule (on some rank)");
            }
            KC_TRY(allgather_rows(c, rows, nunvis, RW, &all, &n_all, nullptr));
            scope.release(visited); scope.release(ulist); scope.release(rows);
            if (n_all == 0) { kc_release(ctx, all); return kc_fail(ctx, KC_E_NO_DEVICE, "fragment ranking stalled without a cycle"); }
            const int CW = 4 + W;
            const u64 cand_cap = n_all + 65536;
            int rc = kc_alloc(ctx, &cand, cand_cap * CW);
            if (rc == KC_OK) {
                KC_LAUNCH((kc_shard_cycle_cand_kernel<W>), kc_grid2(2 * n_all, 256), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, complements, (const u64 *)all, n_all,
                                   cand, cand_cap, cur + 2 * KC_USTRIDE);
                rc = read_words(ctx, cur + 2 * KC_USTRIDE, &ncand, 1);
            }
            u64 over = rc == KC_OK && ncand > cand_cap;
            if (rc == KC_OK) rc = kcc_allreduce_max(c, &over, 1);
            if (rc == KC_OK && over) rc = kc_fail(ctx, KC_E_CAPACITY, "too many candidate edges between cycles and chain ends");
            if (rc == KC_OK) rc = allgather_rows(c, cand, ncand, CW, &allc, &n_allc, nullptr);
            std::vector<u64> h(n_all * RW), hc(n_allc * CW), fix;
            if (rc == KC_OK && hipMemcpyAsync(h.data(), all, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
            if (rc == KC_OK && n_allc && hipMemcpyAsync(hc.data(), allc, hc.size() * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
            (void)hipStreamSynchronize(ctx->stream);
            kc_release(ctx, all); kc_release(ctx, cand); kc_release(ctx, allc);
            KC_TRY(rc);
            const u64 ncyc = kc_cycles_decide<W>(h, n_all, hc, n_allc, fix);
            if (ncyc == 0) return kc_fail(ctx, KC_E_NO_DEVICE, "fragment ranking stalled without a cycle");
            u64 *dfix = nullptr;
            KC_TRY(scope.alloc(&dfix, fix.size()));
            KC_HIP(ctx, hipMemcpyAsync(dfix, fix.data(), fix.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            KC_LAUNCH((kc_shard_apply_fix_kernel<W>), dim3(kc_grid(fix.size() / 2, 256)), dim3(256), 0, ctx->stream, sv, link, (const u64 *)dfix, (u64)(fix.size() / 2), 2 * rlo, 2 * rcap);
            KC_HIP(ctx, hipGetLastError());
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            scope.release(dfix);
            cycles_cut += ncyc;
        }
        scope.release(entries); scope.release(f_exit); scope.release(f_partner); scope.release(f_len); scope.release(f_stream); scope.release(pool); scope.release(chunk_next);
        scope.release(nxt); scope.release(dist); scope.release(last); scope.release(iota); scope.release(act); scope.release(flags8);
    }
    // chain heads gather their tails; winners become chain records
    const int CW = 3 + 3 * W;
    u64 *chains = nullptr, n_chains = 0, covered = 0;
    {
        u64 *heads = nullptr, n_heads = 0;
        KC_TRY(scope.alloc(&heads, n_ent + 1));
        if (n_ent) {
            KC_LAUNCH(kc_shard_head_flags_kernel, dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, (const u64 *)link, (const u64 *)entries, n_ent, flags8);
            KC_TRY(kcu_select_flagged(ctx, iota, flags8, n_ent, heads, cur + 2 * KC_USTRIDE));
            KC_TRY(read_words(ctx, cur + 2 * KC_USTRIDE, &n_heads, 1));
        }
        u64 *req = nullptr, *rq = nullptr, *ans = nullptr, *tail = nullptr;
        u8 *dest = nullptr;
        KC_TRY(scope.alloc(&req, n_heads + 1)); KC_TRY(scope.alloc(&dest, n_heads + 1));
        if (n_heads) KC_LAUNCH(kc_shard_tailreq_kernel, dim3(kc_grid(n_heads, 256)), dim3(256), 0, ctx->stream, (const u64 *)heads, n_heads, (const u64 *)last, req, dest);
        KcRoute rt;
        KC_TRY(kcr_build(c, dest, n_heads, &rt));
        int rc = kcr_forward(c, rt, req, 1, 1, &rq);
        if (rc == KC_OK) rc = kc_alloc(ctx, &ans, rt.n_recv * (W + 1) + 1);
        if (rc == KC_OK && rt.n_recv)
            KC_LAUNCH((kc_shard_tailans_kernel<W>), dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, (const u64 *)rq, rt.n_recv, (const u64 *)f_exit, ans);
        if (rc == KC_OK) rc = kc_alloc(ctx, &tail, n_heads * (W + 1) + 1);
        if (rc == KC_OK) rc = kcr_backward(c, rt, ans, W + 1, tail);
        if (rc == KC_OK) rc = scope.alloc(&chains, (n_heads + 1) * CW);
        KC_HIP(ctx, hipMemsetAsync(cur, 0, 2 * 8, ctx->stream));
        if (rc == KC_OK && n_heads)
            KC_LAUNCH((kc_shard_chain_kernel<W>), dim3(kc_grid(n_heads, 256)), dim3(256), 0, ctx->stream, sv, (const u64 *)link, k, complements, (const u64 *)heads, n_heads,
                               (const u64 *)entries, (const u64 *)dist, (const u64 *)last, (const u64 *)tail, chains, cur);
        u64 hc[2] = {0, 0};
        if (rc == KC_OK) rc = read_words(ctx, cur, hc, 2);
        kcr_free(ctx, &rt);
        kc_release(ctx, rq); kc_release(ctx, ans); kc_release(ctx, tail);
        scope.release(req); scope.release(dest); scope.release(heads);
        KC_TRY(rc);
        n_chains = hc[0]; covered = hc[1];
    }
    {
        u64 tot = covered;
        KC_TRY(kcc_allreduce_sum(c, &tot, 1));
        if (tot != N) return kc_fail(ctx, KC_E_NO_DEVICE, "chain discovery covers " + std::to_string(tot) + " of " + std::to_string(N) + " k-mers");
    }
    // ---------------- the replicated simplitig list
    u64 *all_chains = nullptr, S = 0;
    KC_TRY(allgather_rows(c, chains, n_chains, CW, &all_chains, &S, nullptr));
    scope.release(chains);
    if (S * 5 >= N) { kc_release(ctx, all_chains); return kc_fail_agreed(ctx, KC_E_UNSUPPORTED_SHARDED, "too many simplitigs for the dense overlap graph (main.cpp:175): the sparse branch is not part of the hash-sharded path"); }
    kc_simplitigs *simp = new kc_simplitigs();
    simp->ctx = ctx; simp->k = k; simp->W = W; simp->n = S;
    struct SimpGuard { kc_simplitigs *s; ~SimpGuard() { if (s) kc_simplitigs_destroy(s); } } simpg{simp};
    u64 *fin_sorted = nullptr, *fin_simp = nullptr;
    {
        // order by head key, by the single-GPU flow's own routine (round 6: one MSD bucket pass + LDS sorts where the keys allow it, the
        // stable LSD radix sort otherwise -- the 16 four-bit passes over 62-bit keys were 48 launches, 5 ms of a rank's 39 at its 8-rank
        // share of cfg[3]; the head keys are distinct k-mers, so the order is the same whichever sort makes it)
        u64 *perm = nullptr, *hk = nullptr, *nw = nullptr, *fin = nullptr, *idx = nullptr;
        KC_TRY(scope.alloc(&hk, S * (u64)W + 1));
        KC_TRY(scope.alloc(&nw, S + 1)); KC_TRY(scope.alloc(&fin, S + 1)); KC_TRY(scope.alloc(&idx, S + 1));
        KC_LAUNCH(kc_shard_row_keys_kernel, dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, (const u64 *)all_chains, S, CW, W, hk);
        KC_TRY(kcp_sort_perm_by_keys(ctx, hk, S, W, k, &perm));
        auto perm_back = kc_defer([&] { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, perm); });
        KC_TRY(kc_alloc(ctx, &simp->lens, S + 1)); KC_TRY(kc_alloc(ctx, &simp->woff, S + 1));
        KC_TRY(kc_alloc(ctx, &simp->firstK, S * W + 1)); KC_TRY(kc_alloc(ctx, &simp->lastK, S * W + 1));
        KC_TRY(scope.alloc(&fin_sorted, S + 1)); KC_TRY(scope.alloc(&fin_simp, S + 1));
        KC_LAUNCH(kc_shard_simp_fill_kernel, dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, (const u64 *)all_chains, (const u64 *)perm, S, CW, W, k, simp->lens, nw,
                           simp->firstK, simp->lastK, fin, idx);
        KC_HIP(ctx, hipGetLastError());
        KC_TRY(kcu_exclusive_scan(ctx, nw, simp->woff, S, false));                       // woff[S] = words in total
        // (final fragment id -> simplitig.  The ids are small numbers in 46 bits -- rank and fragment index -- so an MSD bucket pass on their top
        // bits puts everything into one bucket: measured 3.6 ms of hammering one counter before its overflow flag sent it here anyway)
        KC_TRY(kcu_sort_pairs(ctx, fin, fin_sorted, idx, fin_simp, S, 0, 46, false));
        u64 wsum = 0;
        KC_TRY(read_words(ctx, simp->woff + S, &wsum, 1));
        simp->nwords = wsum; simp->total_bases = N + S * (u64)(k - 1);
        KC_TRY(kc_alloc(ctx, &simp->words, wsum + 2));
        KC_HIP(ctx, hipMemsetAsync(simp->words, 0, (wsum + 2) * 8, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        kc_release(ctx, all_chains);
        scope.release(hk); scope.release(nw); scope.release(fin); scope.release(idx);
    }
    KC_HIP(ctx, hipEventRecord(e2, ctx->stream));
    tick("chain records, simplitig list");
    // ---------------- phase 2 (replicated) and the path ranking of phase 3
    KcPathDev path;
    KC_TRY(kco_overlap_path_device(simp, complements, 0, &path, c));
    struct PathGuard { kc_ctx *c; KcPathDev *p; ~PathGuard() { kco_path_release(c, p); } } pathg{ctx, &path};
    KC_HIP(ctx, hipEventRecord(e3, ctx->stream));
    tick("overlap path (phase 2)");
    KcRank rk;
    KC_TRY(kce_rank_path(simp, &path, &rk));
    struct RankGuard { kc_ctx *c; KcRank *r; ~RankGuard() { kce_rank_release(c, r); } } rankg{ctx, &rk};
    const u64 V = path.V, Ltot = rk.L;
    // ---------------- phase 3: bases to the emitting ranks, emit
    u8 *simp_dest = nullptr;
    u64 *rank_rng = nullptr;
    KC_TRY(scope.alloc(&simp_dest, S + 1)); KC_TRY(scope.alloc(&rank_rng, 2 * KC_MAX_RANKS_SH));
    KC_HIP(ctx, hipMemsetAsync(rank_rng, 0xff, KC_MAX_RANKS_SH * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(rank_rng + KC_MAX_RANKS_SH, 0, KC_MAX_RANKS_SH * 8, ctx->stream));
    KC_LAUNCH(kc_shard_simp_dest_kernel, dim3(kc_grid(V, 256)), dim3(256), 0, ctx->stream, (const u64 *)rk.dist, (const u64 *)rk.endv, rk.start, S, V, Ltot, G, simp_dest,
                       rank_rng, rank_rng + KC_MAX_RANKS_SH, (const u64 *)rk.weight);
    u64 n_msg = 0;
    {
        u64 *mcount = nullptr, *moff = nullptr, *msgs = nullptr, *rmsgs = nullptr;
        u8 *mdest = nullptr;
        KC_TRY(scope.alloc(&mcount, n_ent + 2)); KC_TRY(scope.alloc(&moff, n_ent + 2));
        if (n_ent)
            KC_LAUNCH((kc_shard_msg_kernel<false>), dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, k, (const u64 *)entries, n_ent, (const u64 *)f_len, (const u64 *)dist,
                               (const u64 *)last, (const u32 *)f_stream, (const u64 *)pool, (const u32 *)chunk_next, (const u64 *)sv.keys, (const u64 *)link, nstride, set->bshift, W, (const u64 *)fin_sorted,
                               (const u64 *)fin_simp, S, (const u64 *)simp->lens, (const u64 *)simp->woff, (const u8 *)simp_dest, mcount, (const u64 *)nullptr, (u64 *)nullptr, (u8 *)nullptr);
        KC_TRY(kcu_exclusive_scan(ctx, mcount, moff, n_ent));
        KC_TRY(read_words(ctx, moff + n_ent, &n_msg, 1));
        KC_TRY(scope.alloc(&msgs, 2 * n_msg + 2)); KC_TRY(scope.alloc(&mdest, n_msg + 1));
        if (n_ent)
            KC_LAUNCH((kc_shard_msg_kernel<true>), dim3(kc_grid(n_ent, 256)), dim3(256), 0, ctx->stream, k, (const u64 *)entries, n_ent, (const u64 *)f_len, (const u64 *)dist,
                               (const u64 *)last, (const u32 *)f_stream, (const u64 *)pool, (const u32 *)chunk_next, (const u64 *)sv.keys, (const u64 *)link, nstride, set->bshift, W, (const u64 *)fin_sorted,
                               (const u64 *)fin_simp, S, (const u64 *)simp->lens, (const u64 *)simp->woff, (const u8 *)simp_dest, mcount, (const u64 *)moff, msgs, mdest);
        KC_HIP(ctx, hipGetLastError());
        KcRoute rt;
        KC_TRY(kcr_build(c, mdest, n_msg, &rt));
        int rc = kcr_forward(c, rt, msgs, 2, 2, &rmsgs);
        if (rc == KC_OK && rt.n_recv)
            KC_LAUNCH(kc_shard_place_kernel, dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, (const u64 *)rmsgs, rt.n_recv, simp->words);
        (void)hipStreamSynchronize(ctx->stream);
        kcr_free(ctx, &rt);
        kc_release(ctx, rmsgs);
        KC_TRY(rc);
    }
    u64 h_rng[2 * KC_MAX_RANKS_SH];
    KC_TRY(read_words(ctx, rank_rng, h_rng, 2 * KC_MAX_RANKS_SH));
    u64 lo = h_rng[me], hi = h_rng[KC_MAX_RANKS_SH + me];
    if (lo == ~(u64)0) { lo = 0; hi = 0; }   // no block starts in this rank's share
    kc_result *r = new kc_result();
    r->ctx = ctx; r->length = Ltot; r->slice_off = lo; r->slice_len = hi - lo;
    int rc = kc_alloc(ctx, &r->d_out, r->slice_len + 8);
    if (rc == KC_OK && r->slice_len) rc = kce_emit_ranked(simp, &rk, lo, G, me, r->d_out);
    if (rc == KC_OK && want_mask) {
        // -M: the mask probes go to the owners of the k-mers (collective: every rank takes part, whatever its slice holds)
        const u64 n = r->slice_len;
        rc = kc_alloc(ctx, &r->d_mask, n + 8);
        // the first k-1 characters of every slice, to every rank: a k-mer that starts near the end of a slice runs into the next ones
        const int HW = 2 + 16;   // {offset, length, up to 127 characters}
        u64 mine[HW] = {0}, all[64 * HW];
        mine[0] = lo; mine[1] = n;
        if (rc == KC_OK && n) {
            const u64 nh = n < (u64)(k - 1) ? n : (u64)(k - 1);
            if (hipMemcpyAsync(mine + 2, r->d_out, nh, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
        }
        if (rc == KC_OK) rc = kcc_allgather_words(c, mine, all, HW);
        char halo_h[128] = {0};
        if (rc == KC_OK) {
            for (int i = 0; i < k - 1; ++i) {
                const u64 g = hi + (u64)i;   // global position of halo character i
                for (int p = 0; p < G; ++p) {
                    const u64 po = all[p * HW], pn = all[p * HW + 1];
                    if (pn && g >= po && g < po + pn && g - po < (u64)(k - 1)) halo_h[i] = reinterpret_cast<const char *>(all + p * HW + 2)[g - po];
                }
            }
        }
        char *halo = nullptr;
        u64 *tcnt = nullptr, *toff = nullptr, *qkeys = nullptr, *qpos = nullptr, *rq = nullptr, *ans = nullptr, *ansb = nullptr;
        u8 *qdest = nullptr;
        if (rc == KC_OK) rc = scope.alloc(&halo, 128);
        if (rc == KC_OK && hipMemcpyAsync(halo, halo_h, 128, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
        const dim3 gq = kc_grid2(n ? n : 1, 256);
        const u64 ntile = (u64)gq.x * gq.y;
        u64 nqm = 0;
        if (rc == KC_OK) rc = scope.alloc(&tcnt, ntile + 2);
        if (rc == KC_OK) rc = scope.alloc(&toff, ntile + 2);
        if (rc == KC_OK) {
            KC_LAUNCH((kc_shard_mask_query_kernel<W, false>), gq, dim3(256), 0, ctx->stream, (const char *)r->d_out, n, (const char *)halo, lo, Ltot, k, complements, sv, rcapS,
                               r->d_mask, tcnt, (const u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, (u8 *)nullptr);
            rc = kcu_exclusive_scan(ctx, tcnt, toff, ntile);
            if (rc == KC_OK) rc = read_words(ctx, toff + ntile, &nqm, 1);
        }
        if (rc == KC_OK) rc = scope.alloc(&qkeys, nqm * W + 1);
        if (rc == KC_OK) rc = scope.alloc(&qpos, nqm + 1);
        if (rc == KC_OK) rc = scope.alloc(&qdest, nqm + 1);
        if (rc == KC_OK)
            KC_LAUNCH((kc_shard_mask_query_kernel<W, true>), gq, dim3(256), 0, ctx->stream, (const char *)r->d_out, n, (const char *)halo, lo, Ltot, k, complements, sv, rcapS,
                               r->d_mask, tcnt, (const u64 *)toff, qkeys, qpos, qdest);
        KcRoute rt;
        if (rc == KC_OK) rc = kcr_build(c, qdest, nqm, &rt);
        if (rc == KC_OK) rc = kcr_forward(c, rt, qkeys, W, W, &rq);
        if (rc == KC_OK) rc = kc_alloc(ctx, &ans, rt.n_recv + 1);
        if (rc == KC_OK && rt.n_recv)
            KC_LAUNCH((kc_shard_mask_answer_kernel<W>), kc_grid2(rt.n_recv, 256), dim3(256), 0, ctx->stream, sv, (const u64 *)rq, rt.n_recv, ans);
        if (rc == KC_OK) rc = kc_alloc(ctx, &ansb, nqm + 1);
        if (rc == KC_OK) rc = kcr_backward(c, rt, ans, 1, ansb);
        if (rc == KC_OK && nqm)
            KC_LAUNCH(kc_shard_mask_apply_kernel, kc_grid2(nqm, 256), dim3(256), 0, ctx->stream, (const u64 *)qpos, (const u64 *)ansb, nqm, r->d_mask);
        (void)hipStreamSynchronize(ctx->stream);
        kcr_free(ctx, &rt);
        kc_release(ctx, rq); kc_release(ctx, ans); kc_release(ctx, ansb);
    }
    KC_HIP(ctx, hipEventRecord(eend, ctx->stream));
    KC_HIP(ctx, hipEventSynchronize(eend));
    if (rc != KC_OK) { kc_result_destroy(r); return rc; }
    r->stats[0] = N; r->stats[1] = S; r->stats[2] = Ltot; r->stats[4] = V; r->stats[5] = rounds;
    r->stats[6] = (u64)(1000.0 * ev_ms(e0, e1)); r->stats[7] = (u64)(1000.0 * ev_ms(e1, e2));
    r->stats[8] = (u64)(1000.0 * ev_ms(e2, e3)); r->stats[9] = (u64)(1000.0 * ev_ms(e3, eend));
    r->stats[10] = (u64)(1000.0 * ev_ms(ctx->ev[4], ctx->ev[5]));   // neighbour-probe kernel (kc_shard_choose_kernel)
    r->stats[12] = set->cap; r->stats[13] = path.repairs; r->stats[14] = (u64)(1000.0 * ev_ms(e0, eend));
    r->stats_ext[0] = cycles_cut; r->stats_ext[1] = N_local; r->stats_ext[2] = nq;
    tick("path ranking, bases to their ranks, emit");
    if (trace) fprintf(stderr, "[shard %d/%d] entries=%llu jump rounds=%llu chains(local winners)=%llu S=%llu messages=%llu slice=[%llu,%llu) of %llu\n", me, G,
                       (unsigned long long)n_ent, (unsigned long long)rounds, (unsigned long long)n_chains, (unsigned long long)S, (unsigned long long)n_msg,
                       (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)Ltot);
    *out = r;
    return KC_OK;
}


template <int W> int compute_sharded_simplitigs_w(kc_ctx *ctx, kc_comm *c, const u8 *d_seq, u64 len, const u64 *offsets, u64 n_rec, int k, int complements, kc_result **out) {
    KcScope scope(ctx);
    const int G = c->world, me = c->rank;
    hipEvent_t e0 = ctx->ev[0], e3 = ctx->ev[3], eend;
    KC_HIP(ctx, hipEventCreate(&eend));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } evg{eend};
    KC_HIP(ctx, hipEventRecord(e0, ctx->stream));
    // this rank's simplitigs, packed (a record shorter than k or with another letter than ACGT is an error here as on one GPU; the
    // ranks agree on it before anybody enters an exchange)
    kc_simplitigs *mine = nullptr;
    int lrc = KC_OK;
    if (n_rec) lrc = kcp_simplitigs_from_device_ascii(ctx, k, d_seq, len, offsets, n_rec, 1, &mine);
    struct SimpGuard { kc_simplitigs *s; ~SimpGuard() { if (s) kc_simplitigs_destroy(s); } } mineg{mine};
    {
        u64 bad = lrc != KC_OK ? (u64)(-lrc) : 0;
        KC_TRY(kcc_allreduce_max(c, &bad, 1));
        if (bad) return lrc != KC_OK ? lrc : kc_fail(ctx, -(int)bad, "another rank's simplitig records are not valid");
    }
    const int RW = 1 + 2 * W;
    u64 *rows = nullptr, *all = nullptr, S = 0;
    std::vector<u64> counts;
    KC_TRY(scope.alloc(&rows, (n_rec + 1) * RW));
    if (n_rec) KC_LAUNCH(kc_shard_srows_kernel, dim3(kc_grid(n_rec, 256)), dim3(256), 0, ctx->stream, (const u64 *)mine->lens, (const u64 *)mine->firstK, (const u64 *)mine->lastK, n_rec, W, rows);
    KC_HIP(ctx, hipGetLastError());
    KC_TRY(allgather_rows(c, rows, n_rec, RW, &all, &S, &counts));
    scope.release(rows);
    if (S == 0) { kc_release(ctx, all); return kc_fail_agreed(ctx, KC_E_EMPTY, "input cannot be empty"); }
    u64 g0 = 0;
    for (int p = 0; p < me; ++p) g0 += counts[p];
    // the replicated simplitig list (record order: rank by rank)
    kc_simplitigs *simp = new kc_simplitigs();
    simp->ctx = ctx; simp->k = k; simp->W = W; simp->n = S;
    struct SimpGuard2 { kc_simplitigs *s; ~SimpGuard2() { if (s) kc_simplitigs_destroy(s); } } simpg{simp};
    u64 *nw = nullptr;
    KC_TRY(kc_alloc(ctx, &simp->lens, S + 1)); KC_TRY(kc_alloc(ctx, &simp->woff, S + 1));
    KC_TRY(kc_alloc(ctx, &simp->firstK, S * W + 1)); KC_TRY(kc_alloc(ctx, &simp->lastK, S * W + 1));
    KC_TRY(scope.alloc(&nw, S + 1));
    KC_LAUNCH(kc_shard_sfill_kernel, dim3(kc_grid(S, 256)), dim3(256), 0, ctx->stream, (const u64 *)all, S, W, simp->lens, nw, simp->firstK, simp->lastK);
    KC_HIP(ctx, hipGetLastError());
    KC_TRY(kcu_exclusive_scan(ctx, nw, simp->woff, S, false));
    u64 wsum = 0, bases_local = n_rec ? mine->total_bases : 0, bases = 0;
    KC_TRY(read_words(ctx, simp->woff + S, &wsum, 1));
    bases = bases_local;
    KC_TRY(kcc_allreduce_sum(c, &bases, 1));
    simp->nwords = wsum; simp->total_bases = bases;
    KC_TRY(kc_alloc(ctx, &simp->words, wsum + 2));
    KC_HIP(ctx, hipMemsetAsync(simp->words, 0, (wsum + 2) * 8, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, all);
    scope.release(nw);
    const u64 N = bases - S * (u64)(k - 1);
    // phase 2 (the keys of a level divided among the ranks) and the path ranking
    KcPathDev path;
    KC_TRY(kco_overlap_path_device(simp, complements, 0, &path, c));
    struct PathGuard { kc_ctx *c; KcPathDev *p; ~PathGuard() { kco_path_release(c, p); } } pathg{ctx, &path};
    KC_HIP(ctx, hipEventRecord(e3, ctx->stream));
    KcRank rk;
    KC_TRY(kce_rank_path(simp, &path, &rk));
    struct RankGuard { kc_ctx *c; KcRank *r; ~RankGuard() { kce_rank_release(c, r); } } rankg{ctx, &rk};
    const u64 V = path.V, Ltot = rk.L;
    // phase 3: the packed words to the emitting ranks, emit
    u8 *simp_dest = nullptr;
    u64 *rank_rng = nullptr;
    KC_TRY(scope.alloc(&simp_dest, S + 1)); KC_TRY(scope.alloc(&rank_rng, 2 * KC_MAX_RANKS_SH));
    KC_HIP(ctx, hipMemsetAsync(rank_rng, 0xff, KC_MAX_RANKS_SH * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(rank_rng + KC_MAX_RANKS_SH, 0, KC_MAX_RANKS_SH * 8, ctx->stream));
    KC_LAUNCH(kc_shard_simp_dest_kernel, dim3(kc_grid(V, 256)), dim3(256), 0, ctx->stream, (const u64 *)rk.dist, (const u64 *)rk.endv, rk.start, S, V, Ltot, G, simp_dest,
                       rank_rng, rank_rng + KC_MAX_RANKS_SH, (const u64 *)rk.weight);
    {
        const u64 n_msg = n_rec ? mine->nwords : 0;
        u64 *msgs = nullptr, *rmsgs = nullptr;
        u8 *mdest = nullptr;
        KC_TRY(scope.alloc(&msgs, 2 * n_msg + 2)); KC_TRY(scope.alloc(&mdest, n_msg + 1));
        if (n_msg)
            KC_LAUNCH(kc_shard_smsg_kernel, dim3(kc_grid(n_msg, 256)), dim3(256), 0, ctx->stream, (const u64 *)mine->words, (const u64 *)mine->woff, n_rec, n_msg, g0,
                               (const u64 *)simp->woff, (const u8 *)simp_dest, msgs, mdest);
        KC_HIP(ctx, hipGetLastError());
        KcRoute rt;
        KC_TRY(kcr_build(c, mdest, n_msg, &rt));
        int rc = kcr_forward(c, rt, msgs, 2, 2, &rmsgs);
        if (rc == KC_OK && rt.n_recv)
            KC_LAUNCH(kc_shard_place_kernel, dim3(kc_grid(rt.n_recv, 256)), dim3(256), 0, ctx->stream, (const u64 *)rmsgs, rt.n_recv, simp->words);
        (void)hipStreamSynchronize(ctx->stream);
        kcr_free(ctx, &rt);
        kc_release(ctx, rmsgs);
        KC_TRY(rc);
    }
    u64 h_rng[2 * KC_MAX_RANKS_SH];
    KC_TRY(read_words(ctx, rank_rng, h_rng, 2 * KC_MAX_RANKS_SH));
    u64 lo = h_rng[me], hi = h_rng[KC_MAX_RANKS_SH + me];
    if (lo == ~(u64)0) { lo = 0; hi = 0; }   // no block starts in this rank's share
    kc_result *r = new kc_result();
    r->ctx = ctx; r->length = Ltot; r->slice_off = lo; r->slice_len = hi - lo;
    int rc = kc_alloc(ctx, &r->d_out, r->slice_len + 8);
    if (rc == KC_OK && r->slice_len) rc = kce_emit_ranked(simp, &rk, lo, G, me, r->d_out);
    KC_HIP(ctx, hipEventRecord(eend, ctx->stream));
    KC_HIP(ctx, hipEventSynchronize(eend));
    if (rc != KC_OK) { kc_result_destroy(r); return rc; }
    r->stats[0] = N; r->stats[1] = S; r->stats[2] = Ltot; r->stats[4] = V;
    r->stats[8] = (u64)(1000.0 * ev_ms(e0, e3)); r->stats[9] = (u64)(1000.0 * ev_ms(e3, eend));
    r->stats[13] = path.repairs; r->stats[14] = (u64)(1000.0 * ev_ms(e0, eend));
    *out = r;
    return KC_OK;
}

// ---- what the hash-sharded flow does not serve itself: k < 18 (plain hashing, no minimizer blocks), the sparse branch (main.cpp:175-181:
// found out by all ranks together after phase 1), -S with -M (global.h:185-195), -z on table blocks that overflow.  The ranks STAY in
// the collective: their records go to rank 0, which runs the single-GPU path (kc_compute, the same entry point a one-GPU caller uses),
// and every rank receives its slice [L r / G, L (r + 1) / G) of the one result -- the contract of every other sharded call, at the
// speed of one GPU.  (Round 6; until then these requests came back as an error and the CLI ran them on one GPU by itself.)
static int compute_on_root(kc_ctx *ctx, kc_comm *c, const u8 *d_seq, u64 len, const u64 *offsets, u64 n_records, int k, uint32_t flags, kc_result **out) {
    const int G = c->world, me = c->rank;
    const bool simp = (flags & KC_FLAG_SIMPLITIGS_INPUT) != 0, want_mask = (flags & KC_FLAG_WANT_MASK) != 0;
    KcScope scope(ctx);
    u64 mine[2] = {len, n_records}, all[2 * KC_MAX_RANKS_SH];
    KC_TRY(kcc_allgather_words(c, mine, all, 2));
    // rank 0 takes the pieces in rank order: plain input with a '\n' behind every piece (a record boundary: any non-ACGT byte separates
    // k-mers, kmers.h:15-32), simplitig records back to back with their offsets
    u64 total = 0, nrec = 0;
    std::vector<u64> so(G, 0), sc(G, 0), ro(G, 0), rcv(G, 0), piece_at(G, 0), rec_at(G, 0);
    for (int r = 0; r < G; ++r) { piece_at[r] = total; rec_at[r] = nrec; total += all[2 * r] + (simp ? 0 : 1); nrec += all[2 * r + 1]; }
    u8 *gathered = nullptr;
    u64 *d_lens = nullptr, *d_all_lens = nullptr;
    {   // (rank 0's buffer for everybody's records: whether it exists is agreed before anybody enters the exchange)
        u64 bad = 0;
        if (me == 0) {
            if (scope.alloc(&gathered, total + 64) != KC_OK || hipMemsetAsync(gathered, '\n', total + 64, ctx->stream) != hipSuccess) bad = 1;
            for (int r = 0; r < G; ++r) { ro[r] = piece_at[r]; rcv[r] = all[2 * r]; }
        }
        KC_TRY(kcc_allreduce_max(c, &bad, 1));
        if (bad) return kc_fail_agreed(ctx, KC_E_NO_DEVICE, "rank 0 has no device memory for the records of all ranks (a request the hash-sharded flow hands to one GPU)");
    }
    sc[0] = len;
    KC_TRY(kcc_alltoallv(c, d_seq, so.data(), sc.data(), gathered, ro.data(), rcv.data()));
    std::vector<u64> h_offs;
    if (simp) {   // the records' lengths travel the same way (device words), rank 0 turns them back into offsets
        std::vector<u64> lens(n_records + 1, 0);
        for (u64 i = 0; i < n_records; ++i) lens[i] = offsets[i + 1] - offsets[i];
        KC_TRY(scope.alloc(&d_lens, n_records + 1));
        if (n_records) KC_HIP(ctx, hipMemcpyAsync(d_lens, lens.data(), n_records * 8, hipMemcpyHostToDevice, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (lens is a local)
        if (me == 0) {
            KC_TRY(scope.alloc(&d_all_lens, nrec + 1));
            for (int r = 0; r < G; ++r) { ro[r] = rec_at[r] * 8; rcv[r] = all[2 * r + 1] * 8; }
        }
        sc[0] = n_records * 8;
        KC_TRY(kcc_alltoallv(c, d_lens, so.data(), sc.data(), d_all_lens, ro.data(), rcv.data()));
        if (me == 0) {
            h_offs.assign(nrec + 1, 0);
            if (nrec) KC_HIP(ctx, hipMemcpyAsync(h_offs.data() + 1, d_all_lens, nrec * 8, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (u64 i = 0; i < nrec; ++i) h_offs[i + 1] += h_offs[i];
        }
    }
    // rank 0 computes; the others learn the outcome (code, length, statistics) from it
    kc_result *res0 = nullptr;
    u64 head[20] = {0}, heads[20 * KC_MAX_RANKS_SH];
    std::string msg0;
    if (me == 0) {
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int rc0 = kc_compute(ctx, gathered, total, simp ? h_offs.data() : nullptr, simp ? nrec : 0, k, (flags | KC_FLAG_INPUT_ON_DEVICE), &res0);
        if (rc0 != KC_OK) msg0 = ctx->err;
        head[0] = (u64)(long long)rc0;
        if (rc0 == KC_OK) { head[1] = res0->length; for (int i = 0; i < 16; ++i) head[2 + i] = res0->stats[i]; }
    }
    auto drop0 = kc_defer([&] { if (res0) kc_result_destroy(res0); });
    KC_TRY(kcc_allgather_words(c, head, heads, 20));
    const int rc0 = (int)(long long)heads[0];
    if (rc0 != KC_OK) return kc_fail_agreed(ctx, rc0, me == 0 ? msg0 : std::string("rank 0 could not run the request on one GPU (its message says why)"));
    const u64 L = heads[1];
    kc_result *r = new kc_result();
    r->ctx = ctx; r->length = L;
    r->slice_off = L / (u64)G * (u64)me + (L % (u64)G * (u64)me) / (u64)G;   // = floor(L me / G) without a 128-bit product
    {
        const u64 nxt = me + 1 == G ? L : L / (u64)G * (u64)(me + 1) + (L % (u64)G * (u64)(me + 1)) / (u64)G;
        r->slice_len = nxt - r->slice_off;
    }
    for (int i = 0; i < 16; ++i) r->stats[i] = heads[2 + i];
    auto slice_lo = [&](int p) -> u64 { return p >= G ? L : L / (u64)G * (u64)p + (L % (u64)G * (u64)p) / (u64)G; };
    int rc = kc_alloc(ctx, &r->d_out, r->slice_len + 8);
    if (rc == KC_OK && want_mask) rc = kc_alloc(ctx, &r->d_mask, r->slice_len + 8);
    {   // (agreed: a rank that returned alone here would leave its peers waiting in the exchange below)
        u64 bad = rc != KC_OK ? 1 : 0;
        const int rc2 = kcc_allreduce_max(c, &bad, 1);
        if (rc2 != KC_OK || bad) { kc_result_destroy(r); return rc2 != KC_OK ? rc2 : kc_fail_agreed(ctx, KC_E_NO_DEVICE, "no device memory for a slice of the result (on some rank)"); }
    }
    for (int which = 0; which < (want_mask ? 2 : 1); ++which) {
        for (int p = 0; p < G; ++p) { so[p] = 0; sc[p] = 0; ro[p] = 0; rcv[p] = 0; }
        if (me == 0) for (int p = 0; p < G; ++p) { so[p] = slice_lo(p); sc[p] = slice_lo(p + 1) - slice_lo(p); }
        rcv[0] = r->slice_len;
        const void *src = me == 0 ? (which ? (const void *)res0->d_mask : (const void *)res0->d_out) : nullptr;
        rc = kcc_alltoallv(c, src, so.data(), sc.data(), which ? (void *)r->d_mask : (void *)r->d_out, ro.data(), rcv.data());
        if (rc != KC_OK) { kc_result_destroy(r); return rc; }
    }
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = r;
    return KC_OK;
}

}  // namespace

extern "C" int kc_compute_sharded(kc_ctx *ctx, kc_comm *comm, const void *seq, uint64_t len, int k, uint32_t flags, kc_result **out) {
    if (!ctx || !comm || !out || (!seq && len)) return KC_E_ARG;
    *out = nullptr;
    ctx->err_agreed = false;
    if (comm->ctx != ctx) return kc_fail(ctx, KC_E_ARG, "the communicator belongs to another context");
    if (k < 1 || k > 127) return kc_fail_agreed(ctx, KC_E_ARG, "k must be in 1..127");
    if (flags & KC_FLAG_SIMPLITIGS_INPUT) return kc_fail_agreed(ctx, KC_E_ARG, "-S input goes through kc_compute_sharded_simplitigs (records + offsets)");
    const int min_frequency = ((flags >> 8) & 0xff) ? (int)((flags >> 8) & 0xff) : 1;   // -z (parser.h:122-141): counted where all occurrences of a k-mer meet, at its owner
    const int want_mask = (flags & KC_FLAG_WANT_MASK) ? 1 : 0;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const int complements = (flags & KC_FLAG_COMPLEMENTS) ? 1 : 0;
    u8 *d_own = nullptr;
    const u8 *d_seq = (const u8 *)seq;
    if (!(flags & KC_FLAG_INPUT_ON_DEVICE)) {
        KC_TRY(kc_alloc(ctx, &d_own, len + 64));
        if (len) KC_HIP(ctx, hipMemcpyAsync(d_own, seq, len, hipMemcpyHostToDevice, ctx->stream));
        d_seq = d_own;
    }
    int rc = KC_E_UNSUPPORTED_SHARDED;
    // KC_SHARD_NO_ROOT_FALLBACK=1 (test knob): the requests the sharded flow does not serve come back as KC_E_UNSUPPORTED_SHARDED
    const bool root_ok = !getenv("KC_SHARD_NO_ROOT_FALLBACK");
    if (k < 18) ctx->err_agreed = true;   // (k < 18: plain hashing, no minimizer blocks to shard by -- known to every rank without an exchange)
    else switch (kc_words_for_k(k)) {
        case 1: rc = compute_sharded_w<1>(ctx, comm, d_seq, len, k, complements, want_mask, min_frequency, out); break;
        case 2: rc = compute_sharded_w<2>(ctx, comm, d_seq, len, k, complements, want_mask, min_frequency, out); break;
        default: rc = compute_sharded_w<4>(ctx, comm, d_seq, len, k, complements, want_mask, min_frequency, out); break;
    }
    // every rank holds the same KC_E_UNSUPPORTED_SHARDED here (agreed errors are taken together, before any further exchange): rank 0
    // runs the request, everybody gets its slice
    if (rc == KC_E_UNSUPPORTED_SHARDED && ctx->err_agreed) {
        if (root_ok) { (void)hipStreamSynchronize(ctx->stream); ctx->err_agreed = false; rc = compute_on_root(ctx, comm, d_seq, len, nullptr, 0, k, flags, out); }
        else if (k < 18) rc = kc_fail_agreed(ctx, KC_E_UNSUPPORTED_SHARDED, "k < 18 (plain hashing, no minimizer blocks) is not part of the hash-sharded path");
    }
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d_own);
    kcc_after_collective_call(comm, rc);
    return rc;
}

// -S on the sharded path: `seq` = this rank's simplitig records one after the other (no separators), offsets[0..n_records] their
// starts and the end, as kc_compute takes them (simplitigs_from_fasta, simplitigs.h:89-103).  Collective.
extern "C" int kc_compute_sharded_simplitigs(kc_ctx *ctx, kc_comm *comm, const void *seq, uint64_t len, const uint64_t *offsets, uint64_t n_records, int k, uint32_t flags,
                                             kc_result **out) {
    if (!ctx || !comm || !out || (!seq && len) || (!offsets && n_records)) return KC_E_ARG;
    *out = nullptr;
    ctx->err_agreed = false;
    if (comm->ctx != ctx) return kc_fail(ctx, KC_E_ARG, "the communicator belongs to another context");
    if (k < 1 || k > 127) return kc_fail_agreed(ctx, KC_E_ARG, "k must be in 1..127");
    if (((flags >> 8) & 0xff) > 1) return kc_fail(ctx, KC_E_ARG, "Inputting simplitigs is not compatible with frequency filterring.");
    if ((flags & KC_FLAG_WANT_MASK) && getenv("KC_SHARD_NO_ROOT_FALLBACK")) return kc_fail_agreed(ctx, KC_E_UNSUPPORTED_SHARDED, "-S with -M is not part of the hash-sharded path");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const int complements = (flags & KC_FLAG_COMPLEMENTS) ? 1 : 0;
    u8 *d_own = nullptr;
    const u8 *d_seq = (const u8 *)seq;
    if (!(flags & KC_FLAG_INPUT_ON_DEVICE)) {
        KC_TRY(kc_alloc(ctx, &d_own, len + 64));
        if (len) KC_HIP(ctx, hipMemcpyAsync(d_own, seq, len, hipMemcpyHostToDevice, ctx->stream));
        d_seq = d_own;
    }
    int rc;
    // -S with -M (global.h:185-195: the mask needs the k-mer set of all simplitigs): rank 0 runs it, everybody gets its slice
    if (flags & KC_FLAG_WANT_MASK) rc = compute_on_root(ctx, comm, d_seq, len, offsets, n_records, k, flags | KC_FLAG_SIMPLITIGS_INPUT, out);
    else switch (kc_words_for_k(k)) {
        case 1: rc = compute_sharded_simplitigs_w<1>(ctx, comm, d_seq, len, offsets, n_records, k, complements, out); break;
        case 2: rc = compute_sharded_simplitigs_w<2>(ctx, comm, d_seq, len, offsets, n_records, k, complements, out); break;
        default: rc = compute_sharded_simplitigs_w<4>(ctx, comm, d_seq, len, offsets, n_records, k, complements, out); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d_own);
    kcc_after_collective_call(comm, rc);
    return rc;
}
