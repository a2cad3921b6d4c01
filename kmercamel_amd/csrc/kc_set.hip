// Phase 0 of the BIGREEDY path on gfx950: ASCII -> packed 2-bit sequence, rolling canonical k-mers, open-addressing
// set build in HBM.  Replaces AddKMers + kh_put (parser.h:22-49, khash.h:311-352) and the khash set itself
// (khash.h:215-359, khash_utils.h:28-95).  Integer/byte work, HBM-bound: no LDS tiling is needed for the packed
// stream (each thread reads its own few words, neighbouring lanes read neighbouring words), the cost is the random
// 8-byte (W*8-byte) slot access of the insert.
#include "kc_internal.h"
#include "kc_comm.h"
#include "kc_scan_dev.h"
#include <chrono>

// ------------------------------------------------------------------------------------------------- pool / ctx
hipError_t KcPool::alloc(void **out, size_t bytes) {
    const size_t gran = (size_t)2 << 20;
    size_t need = (bytes + gran - 1) / gran * gran;
    int best = -1;
    for (size_t i = 0; i < free_list.size(); ++i) {
        if (free_list[i].size >= need && free_list[i].size <= need * 2 + gran) {
            if (best < 0 || free_list[i].size < free_list[best].size) best = (int)i;
        }
    }
    if (best >= 0) {
        *out = free_list[best].p;
        live[*out] = free_list[best].size;
        live_bytes += free_list[best].size;
        if (getenv("KC_POOL_TRACE") && need >= ((size_t)1 << 30)) fprintf(stderr, "[pool] reuse %.2f GB for %.2f GB -> live %.2f GB\n", free_list[best].size / 1e9, need / 1e9, live_bytes / 1e9);
        free_list.erase(free_list.begin() + best);
        if (live_bytes > peak_bytes) peak_bytes = live_bytes;
        return hipSuccess;
    }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, need);
    malloc_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    malloc_bytes += need; ++malloc_calls;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim();
        e = hipMalloc(out, need);
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
    }
    live[*out] = need;
    live_bytes += need;
    if (live_bytes > peak_bytes) peak_bytes = live_bytes;
    static const bool trace = getenv("KC_POOL_TRACE") != nullptr;
    if (trace && need >= ((size_t)1 << 30)) fprintf(stderr, "[pool] hipMalloc %.2f GB -> live %.2f GB (cached %zu blocks)\n", need / 1e9, live_bytes / 1e9, free_list.size());
    return hipSuccess;
}
void KcPool::release(void *p) {
    auto it = live.find(p);
    if (it == live.end()) return;
    free_list.push_back({it->second, p});
    live_bytes -= it->second;
    live.erase(it);
}
void KcPool::trim() {
    for (auto &b : free_list) (void)hipFree(b.p);
    free_list.clear();
}
KcPool::~KcPool() {
    trim();
    for (auto &kv : live) (void)hipFree(kv.first);
    live.clear();
}

extern "C" int kc_ctx_create(int device, kc_ctx **out) {
    if (!out) return KC_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return KC_E_NO_DEVICE; }
    if (device < 0 || device >= ndev) return KC_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return KC_E_NO_DEVICE;
    kc_ctx *ctx = new kc_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return KC_E_NO_DEVICE; }
    for (int i = 0; i < 8; ++i) {
        if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return KC_E_NO_DEVICE; }
    }
    *out = ctx;
    return KC_OK;
}
extern "C" void kc_ctx_destroy(kc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    kc_pin_free_all(ctx);
    for (int i = 0; i < 8; ++i) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" const char *kc_last_error(const kc_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }
extern "C" void *kc_ctx_stream(kc_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" int kc_ctx_synchronize(kc_ctx *ctx) {
    if (!ctx) return KC_E_ARG;
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}
extern "C" int kc_ctx_stats(kc_ctx *ctx, uint64_t *stats) {
    if (!ctx || !stats) return KC_E_ARG;
    stats[0] = ctx->pool.peak_bytes; stats[1] = ctx->pool.malloc_bytes; stats[2] = ctx->pool.malloc_calls;
    stats[3] = (u64)(ctx->pool.malloc_seconds * 1e6);
    return KC_OK;
}
extern "C" void kc_free(void *p) { free(p); }
extern "C" int kc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
extern "C" const char *kc_version(void) { return "kmercamel_amd 0.1 (gfx950; path of KmerCamel v2.3.0 compute/greedy)"; }

// ------------------------------------------------------------------------------------------------- pack2bit
// Per 64 input bytes: two 2-bit code words (32 bases each, first base in bits 63:62), one break-bit word (bit j = byte j is not
// ACGTacgt, kmers.h:15-32) and one lower-case-bit word (bit j = byte > 'Z', parser.h:41).  A lane works on 16-byte pieces, lane after
// lane (four per lane, 1 KB per load instruction of a wavefront), and writes each piece's 32 code bits and its two 16-bit masks where they
// belong inside those words: nothing crosses lanes.  (A thread that read its own 64 bytes in four loads had the wavefront's lanes 64 bytes
// apart in every one of them, and the lines came from L2 up to four times: 1.2 ms for 2.6 GB.)
__global__ void __launch_bounds__(256) kc_pack2bit_kernel(const u8 *__restrict__ seq, u64 len, u64 *__restrict__ codes,
                                                          u64 *__restrict__ brk, u64 *__restrict__ low, u64 nblk) {
    const u64 T = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 q0 = (T >> 6) * 256 + (T & 63), npieces = nblk * 4;
    const bool aligned = (((uintptr_t)seq) & 15) == 0;
    u32 *codes32 = reinterpret_cast<u32 *>(codes);
    unsigned short *brk16 = reinterpret_cast<unsigned short *>(brk), *low16 = reinterpret_cast<unsigned short *>(low);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 q = q0 + (u64)i * 64;
        if (q >= npieces) break;
        const u64 base = q * 16;
        u32 vv[4];
        if (aligned && base + 16 <= len) {
            const uint4 v = *reinterpret_cast<const uint4 *>(seq + base);
            vv[0] = v.x; vv[1] = v.y; vv[2] = v.z; vv[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 w = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) w |= (u32)((base + (u64)(4 * j + r) < len) ? seq[base + (u64)(4 * j + r)] : (u8)'\n') << (8 * r);
                vv[j] = w;
            }
        }
        u32 c = 0, bm = 0, lm = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 ch = (vv[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const u32 up = ch & 0xDFu;
            const bool ok = (up == 'A') | (up == 'C') | (up == 'G') | (up == 'T');
            u32 code = (up >> 1) & 3u;   // A0 C1 G3 T2
            code ^= code >> 1;           // A0 C1 G2 T3
            c |= code << (30 - 2 * j);
            bm |= (u32)(!ok) << j;
            lm |= (u32)(ch > (u32)'Z') << j;
        }
        codes32[q ^ 1] = c;   // (the word's first 16 bases are its high half)
        brk16[q] = (unsigned short)bm;
        low16[q] = (unsigned short)lm;
    }
}

int kcs_pack(kc_ctx *ctx, const void *d_seq, u64 len, u64 **d_codes, u64 **d_brk, u64 **d_low) {
    u64 nblk = (len + 63) / 64;
    if (nblk == 0) nblk = 1;
    KC_TRY(kc_alloc(ctx, d_codes, nblk * 2 + 2));
    KC_TRY(kc_alloc(ctx, d_brk, nblk + 1));
    KC_TRY(kc_alloc(ctx, d_low, nblk + 1));
    KC_LAUNCH(kc_pack2bit_kernel, dim3(kc_grid((nblk + 63) / 64 * 64, 256)), dim3(256), 0, ctx->stream, (const u8 *)d_seq, len, *d_codes,
                       *d_brk, *d_low, nblk);
    KC_HIP(ctx, hipGetLastError());
    return KC_OK;
}

// ------------------------------------------------------------------------------------------------- extract + insert
// AddKMers (parser.h:22-49) over the packed stream: each thread owns PPT consecutive k-mer end positions, warms the
// rolling forward / reverse-complement windows up over the k-1 preceding bases, then rolls (parser.h:38-39) and
// inserts the canonical k-mer (parser.h:44) of every window that holds k valid bases (and, if case_sensitive, whose
// first base is upper case, parser.h:41-42).
// kc_roll_kmers walks one thread's window run and hands every k-mer to `emit(key, minimizer hash, first position)`.
static inline __host__ __device__ int kc_roll_win(int k, int m) { return (m && k - m - 1 >= 1) ? k - m - 1 : 1; }
template <int W, typename F>
__device__ __forceinline__ void kc_roll_kmers(const u64 *__restrict__ codes, const u64 *__restrict__ brk, const u64 *__restrict__ low, u64 len,
                                              int k, int complements, int case_sensitive, int m, int ppt, u32 *ring, F &&emit) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 e0 = t * (u64)ppt;
    if (e0 >= len) return;
    u64 e1 = e0 + (u64)ppt;
    if (e1 > len) e1 = len;
    u64 s = e0 >= (u64)(k - 1) ? e0 - (u64)(k - 1) : 0;
    Key<W> cur = key_zero<W>(), rc = key_zero<W>();
    int run = 0;
    // minimizer of the k-mer's MIDDLE (positions 1..k-2; kc_set_dev.h) = minimum of the win = k-m-1 canonical m-mer hashes that end
    // before the current base (one new hash per base).  Sliding minimum in O(1) per
    // base (van Herk / Gil-Werman): the positions are cut into blocks of `win`; a window is a suffix of the previous block
    // plus a prefix of the current one, so it is min(suffix minimum stored when that block ended, running prefix minimum).
    // Per thread, in LDS (index j * blockDim + tid, conflict-free): raw[win] hashes of the current block, suf[win] suffix
    // minima of the previous one.  (For k = 127 the window has 113 hashes: scanning it per k-mer was most of the kernel.)
    const int win = kc_roll_win(k, m);
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    u32 *raw = ring, *suf = ring + (size_t)win * blockDim.x;
    u32 pref = ~0u;
    u32 prev_min = ~0u;   // minimum of the `win` m-mer hashes that end before the current base = those inside the middle of the k-mer ending here
    u64 cw = codes[s >> 5], bw = brk[s >> 6];
    u32 rj = 0;   // position inside the current block (a running counter: p % win would be a 64-bit division per base)
    for (u64 p = s; p < e1; ++p, rj = (rj + 1 == (u32)win) ? 0u : rj + 1) {
        if ((p & 31) == 0) cw = codes[p >> 5];
        if ((p & 63) == 0) bw = brk[p >> 6];
        u64 b = (cw >> (62 - 2 * (p & 31))) & 3;
        if (rj == 0) pref = ~0u;
        if ((bw >> (p & 63)) & 1) run = 0;
        else {
            cur = key_append(cur, b, k);
            rc = key_prepend(rc, 3 - b, k);
            if (run < k) ++run;
            if (p >= e0 && run >= k) {
                bool keep = true;
                if (case_sensitive) {
                    u64 fp = p - (u64)(k - 1);
                    keep = !((low[fp >> 6] >> (fp & 63)) & 1);
                }
                if (keep) {
                    const Key<W> &key = (!complements || key_lt(cur, rc)) ? cur : rc;
                    emit(key, prev_min);
                }
            }
            if (m && run >= m) {
                const u32 h = kc_mmer_hash((u32)cur.w[0] & maskm, (u32)key_shr(rc, 2 * (k - m)).w[0]);
                raw[rj * blockDim.x + threadIdx.x] = h;
                pref = h < pref ? h : pref;
            }
        }
        if (m) {
            // window minimum of the `win` hashes ending HERE: prefix of the current block, suffix of the previous one
            prev_min = pref;
            if (rj + 1 < (u32)win) { const u32 sv = suf[(rj + 1) * blockDim.x + threadIdx.x]; prev_min = sv < prev_min ? sv : prev_min; }
            if (rj + 1 == (u32)win) {   // block complete: its suffix minima serve the windows of the next block
                u32 acc = ~0u;
                for (int i = win - 1; i >= 0; --i) {
                    const u32 h = raw[i * blockDim.x + threadIdx.x];
                    acc = h < acc ? h : acc;
                    suf[i * blockDim.x + threadIdx.x] = acc;
                }
            }
        }
    }
}

template <int W>
__global__ void __launch_bounds__(256) kc_insert_kernel(const u64 *__restrict__ codes, const u64 *__restrict__ brk,
                                                        const u64 *__restrict__ low, u64 len, int k, int complements,
                                                        int case_sensitive, SetView<W> set, u64 *counters, int ppt, u32 *occ_counts,
                                                        const u64 *__restrict__ only_over, u64 over_limit) {
    // only_over != nullptr: the clean-up pass after a bucketed build whose buckets overflowed -- only the k-mers of blocks
    // that received more than over_limit occurrences (only_over[block], the scatter kernel's cursors) are inserted
    extern __shared__ u32 kc_ring[];
    u64 added = 0;
    bool overflow = false;
    const int m = set.m;
    kc_roll_kmers<W>(codes, brk, low, len, k, complements, case_sensitive, m, ppt, kc_ring, [&](const Key<W> &key, u32 mmh) {
        const KcPlace pl = m ? set_place_mmh(set, key, mmh) : set_place(set, key);
        if (only_over && only_over[pl.home >> set.bshift] <= over_limit) return;
        u64 slot;
        int r = set_insert_at(set, key, pl, &slot);
        if (occ_counts && r >= 0) atomicAdd(&occ_counts[slot], 1u);   // AddKMersWithFrequencies (parser.h:51-85)
        if (r > 0) ++added;
        else if (r < 0) overflow = true;
    });
    if (added) atomicAdd((ull *)&counters[0], (ull)added);
    if (overflow) atomicExch((ull *)&counters[1], 1ULL);
}

// ---- bucketed build (single-word keys): global CAS inserts pay a fabric atomic plus two scattered line round trips per
// k-mer and were the slowest kernel of the path.  Instead (1) scatter the canonical k-mers into one bucket per table
// block -- consecutive k-mers share their minimizer, so a thread reserves space once per RUN of equal blocks; (2) one
// workgroup per block builds that block's table (16384 / 8192 / 4096 slots for 1 / 2 / 4 key words) in LDS (LDS compare-and-swap, no global atomics) and writes it
// out with coalesced stores; (3) the few keys whose 128 local probes were all taken go through the global insert.
// (positions per thread: 16 single-word keys, 8 wide ones -- the keys of an open run are staged in LDS)
template <int W>
__global__ void __launch_bounds__(256) kc_bucket_scatter_kernel(const u64 *__restrict__ codes, const u64 *__restrict__ brk,
                                                                const u64 *__restrict__ low, u64 len, int k, int complements,
                                                                int case_sensitive, SetView<W> set, u64 *bucket, u64 *bcursor, u64 *counters,
                                                                int ppt, u32 stage_cap, const u64 *__restrict__ boff, int count_only) {
    // boff == nullptr: one bucket of 2^bshift entries per block (single GPU).  Otherwise (hash-sharded build) the kernel runs
    // twice: count_only = 1 adds the run lengths to bcursor[block]; then, with boff = exclusive scan of those counts and the
    // cursors cleared, the keys land at boff[block] + position -- grouped by block, hence contiguous per owner rank.
    extern __shared__ u32 kc_ring[];
    const int win = kc_roll_win(k, set.m);
    u64 *stage = reinterpret_cast<u64 *>(kc_ring + (size_t)2 * win * blockDim.x);   // stage[(j * W + w) * blockDim + tid]: keys of the open run
    const u64 bs = (u64)1 << set.bshift;
    u32 nrun = 0;
    u64 run_blk = 0;
    bool overflow = false;
    auto flush = [&]() {
        if (!nrun) return;
        if (count_only) { atomicAdd((ull *)&bcursor[run_blk], (ull)nrun); nrun = 0; return; }
        const u64 base = atomicAdd((ull *)&bcursor[run_blk], (ull)nrun);
        if (boff) {
            u64 *dst = bucket + (boff[run_blk] + base) * W;
            for (u32 j = 0; j < nrun * W; ++j) dst[j] = stage[j * blockDim.x + threadIdx.x];
            nrun = 0;
            return;
        }
        // a bucket that overflows keeps the part of the run that still fits (so that its first `bs` entries are all real
        // keys for the build kernel) and raises the flag: the caller then re-inserts the whole input through the global path
        u32 fit = nrun;
        if (base + nrun > bs) { overflow = true; fit = base < bs ? (u32)(bs - base) : 0; }
        u64 *dst = bucket + ((run_blk << set.bshift) + base) * W;
        for (u32 j = 0; j < fit * W; ++j) dst[j] = stage[j * blockDim.x + threadIdx.x];
        nrun = 0;
    };
    kc_roll_kmers<W>(codes, brk, low, len, k, complements, case_sensitive, set.m, ppt, kc_ring, [&](const Key<W> &key, u32 mmh) {
        const u64 blk = set_block_home(set, mmh, 0) >> set.bshift;
        if (nrun && (blk != run_blk || nrun == stage_cap)) flush();   // (a run longer than the staging area is reserved in pieces)
        run_blk = blk;
        if (!count_only)
            for (int w = 0; w < W; ++w) stage[(nrun * W + w) * blockDim.x + threadIdx.x] = key.w[w];
        ++nrun;
    });
    flush();
    if (overflow) atomicExch((ull *)&counters[3], 1ULL);
}

// ---- the same scatter, wave-cooperative (single-word keys, minimizer tables).  The rolling form above keeps two rings of
// win = k - m - 1 hashes and the keys of the open run in LDS for every thread (176 B per thread: 3 waves per SIMD), rolls k - 1
// warm-up bases per thread and waits for every reservation where it is issued.  Here a LANE IS A POSITION: lane i of a wave takes
// the k-mer that ends at position p0 + i straight out of the packed sequence (two words and a funnel shift -- no rolling, no
// warm-up), hashes the m-mer that ends there, and the minimizer of the k-mer's middle is a sliding minimum over the win lanes
// before it: doubling steps with shuffles, the lanes below 0 come from the previous chunk's registers (a wave walks `chunks`
// consecutive 64-position chunks, one extra in front for those).  Runs of equal blocks are found with a ballot; the first lane
// of a run reserves for all of it, the others get the base by shuffle and store their own key.  No LDS at all.
// A key that finds its block's bucket full goes to a side list (counters[5] counts them; they are inserted like the keys that find no
// room in their block's probe run).  Called by ALL lanes of a wave (`need`: this lane has such a key); returns true for a lane whose key
// was NOT saved -- no list, or the list is full: the caller then raises the flag that sends the blocks concerned through a second sweep
// of the whole input.  One reservation per wave and call, none once the wave has seen the list full (`full`, wave-uniform): a repeat that
// sends 10^7 occurrences to one block must not turn into as many same-address atomics.  (Round 5: one overflowing bucket of 10^8
// 127-mers -- block sizes vary by a factor of two there -- had cost 15 ms of a 40 ms pass.)
template <int W> __device__ __forceinline__ bool kc_scatter_aside(bool need, const u64 *key, u64 *sovf, u64 sovf_cap, u64 *counters, bool &full, int lane) {
    const unsigned long long m = __ballot(need);
    if (!m) return false;
    if (!sovf || full) return need;
    const int leader = __builtin_ctzll(m);
    const u64 n = (u64)__popcll(m);
    u64 base = 0;
    if (lane == leader) base = atomicAdd((ull *)&counters[5], (ull)n);
    base = __shfl(base, leader, 64);
    if (base + n >= sovf_cap) full = true;
    if (!need) return false;
    const u64 pos = base + (u64)__popcll(m & (((unsigned long long)1 << lane) - 1));
    if (pos >= sovf_cap) return true;
    for (int w = 0; w < W; ++w) sovf[pos * W + w] = key[w];
    return false;
}
__device__ __forceinline__ u64 kc_extract_bases(const u64 *__restrict__ codes, u64 s, int nb) {   // the nb <= 32 bases from position s on, right-aligned
    const u64 wi = s >> 5;
    const int o = (int)(s & 31);
    u64 v = codes[wi] << (2 * o);
    if (o && 32 - o < nb) v |= codes[wi + 1] >> (64 - 2 * o);
    return v >> (64 - 2 * nb);
}
__device__ __forceinline__ bool kc_any_break(const u64 *__restrict__ brk, u64 s, int nb) {         // a break bit among the nb <= 32 positions from s on
    const u64 wi = s >> 6;
    const int o = (int)(s & 63);
    u64 b = brk[wi] >> o;
    if (o + nb > 64) b |= brk[wi + 1] << (64 - o);
    return (b & (((u64)1 << nb) - 1)) != 0;
}
__device__ __forceinline__ u32 kc_halo_get(u32 prev, u32 cur, int idx) {   // value of lane idx of this chunk, or of lane 64 + idx of the previous one
    const u32 a = __shfl(cur, idx & 63, 64), b = __shfl(prev, idx & 63, 64);
    return idx >= 0 ? a : b;
}
__global__ void __launch_bounds__(256) kc_bucket_scatter_wave_kernel(const u64 *__restrict__ codes, const u64 *__restrict__ brk, const u64 *__restrict__ low, u64 len, int k,
                                                                     int complements, int case_sensitive, SetView<1> set, u64 *bucket, u64 *bcursor, u64 *counters,
                                                                     int chunks, u64 *sovf = nullptr, u64 sovf_cap = 0) {
    const int lane = threadIdx.x & 63;
    const int exp_ = chunks >> 16;   // (timing experiments only: 1 no atomics, 2 no stores, 3 neither -- tools/r5_build_ab.py; results are wrong then)
    chunks &= 0xffff;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 P0 = wave * (u64)chunks * 64;
    if (P0 >= len) return;
    const int m = set.m, win = k - m - 1;
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const u64 bs = (u64)1 << set.bshift;
    u32 a1p = ~0u, a2p = ~0u, a4p = ~0u, a8p = ~0u;
    bool overflow = false, aside_full = false;
    struct Chunk { bool valid, starts; u64 key, blk; unsigned long long vm, sm; };
    // one chunk: this lane's k-mer and block, the wave's run structure (c = -1: only the hashes, for the halo of chunk 0)
    auto compute = [&](int c) -> Chunk {
        Chunk R{false, false, 0, ~(u64)0, 0, 0};
        const long long pp = (long long)P0 + 64LL * c + lane;
        const bool inr = pp >= 0 && (u64)pp < len;
        const u64 p = inr ? (u64)pp : 0;
        // canonical hash of the m-mer that ends here (all ones: none)
        u32 h = ~0u;
        if (inr && p + 1 >= (u64)m && !kc_any_break(brk, p + 1 - (u64)m, m)) {
            const u32 mm = (u32)kc_extract_bases(codes, p + 1 - (u64)m, m);
            u32 r = __brev(~mm);
            r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            r >>= 32 - 2 * m;
            h = kc_mmer_hash(mm & maskm, r);
        }
        // the k-mer that ends here
        if (c >= 0 && inr && p + 1 >= (u64)k && !kc_any_break(brk, p + 1 - (u64)k, k)) {
            const u64 fp = p + 1 - (u64)k;
            R.valid = !case_sensitive || !((low[fp >> 6] >> (fp & 63)) & 1);
            if (R.valid) {
                Key<1> f; f.w[0] = kc_extract_bases(codes, fp, k);
                R.key = f.w[0];
                if (complements) { const Key<1> r = key_rc(f, k); if (r.w[0] < R.key) R.key = r.w[0]; }
            }
        }
        // sliding minimum over the win lanes before this one
        const u32 a1 = h;
        u32 t = kc_halo_get(a1p, a1, lane - 1); const u32 a2 = a1 < t ? a1 : t;
        t = kc_halo_get(a2p, a2, lane - 2); const u32 a4 = a2 < t ? a2 : t;
        t = kc_halo_get(a4p, a4, lane - 4); const u32 a8 = a4 < t ? a4 : t;
        u32 pm = ~0u;
        int at = lane - 1;
        if (win & 8) { t = kc_halo_get(a8p, a8, at); pm = t < pm ? t : pm; at -= 8; }
        if (win & 4) { t = kc_halo_get(a4p, a4, at); pm = t < pm ? t : pm; at -= 4; }
        if (win & 2) { t = kc_halo_get(a2p, a2, at); pm = t < pm ? t : pm; at -= 2; }
        if (win & 1) { t = kc_halo_get(a1p, a1, at); pm = t < pm ? t : pm; }
        a1p = a1; a2p = a2; a4p = a4; a8p = a8;
        // runs of equal blocks among the valid lanes
        R.blk = R.valid ? set_block_home(set, pm, 0) >> set.bshift : ~(u64)0;
        const u64 blk_before = __shfl_up(R.blk, 1, 64);
        R.starts = R.valid && (lane == 0 || blk_before != R.blk);
        R.vm = __ballot(R.valid); R.sm = __ballot(R.starts);
        return R;
    };
    // the first lane of a run reserves for all of it ...
    auto reserve = [&](const Chunk &R) -> u64 {
        if (!R.starts) return 0;
        const unsigned long long stop = (R.sm | ~R.vm) >> lane >> 1;   // the run ends before the next start or invalid lane (two shifts: lane 63 shifts by 64 in all)
        const int rl = stop ? __builtin_ctzll(stop) + 1 : 64 - lane;
        if (exp_ & 1) return (kc_mix64(P0 + (u64)lane) >> 20) % (bs - 64);
        return atomicAdd((ull *)&bcursor[R.blk], (ull)rl);
    };
    // ... the others get the base by shuffle and store their own key
    auto commit = [&](const Chunk &R, u64 base) {
        const unsigned long long below = R.sm & (((unsigned long long)2 << lane) - 1);   // the starts at or below this lane
        const int leader = below ? 63 - __builtin_clzll(below) : 0;
        base = __shfl(base, leader, 64);
        const u64 at2 = base + (u64)(lane - leader);
        if (R.valid) {
            if (exp_ & 2) { if (at2 == ~(u64)0) bucket[0] = R.key; }
            else if (at2 < bs) bucket[(R.blk << set.bshift) + at2] = R.key;
        }
        if (kc_scatter_aside<1>(R.valid && !(exp_ & 2) && at2 >= bs, &R.key, sovf, sovf_cap, counters, aside_full, lane)) overflow = true;
    };
    (void)compute(-1);
    // two chunks per trip: both reservations are in flight before either is waited for
    for (int c = 0; c < chunks; c += 2) {
        if (P0 + 64ULL * (u64)c >= len) break;   // (wave-uniform)
        const Chunk A = compute(c), B = compute(c + 1);
        const u64 ba = reserve(A), bb = reserve(B);
        commit(A, ba);
        commit(B, bb);
    }
    if (overflow) atomicExch((ull *)&counters[3], 1ULL);
}

// ---- round 5: the wave-cooperative scatter again, leaner.  Measured on the form above at 2.5 * 10^9 31-mers (tools/r5_build_ab.sh,
// KC_SCATTER_EXP): 24.5 ms as it is, 24.0 without the reservations, 20.5 without the stores, 18.2 with neither -- the kernel is its own
// arithmetic and the waits for its per-lane loads (PMC: 174 vector + 76 scalar instructions per chunk of 64 positions, waves waiting 74 % of
// their time: four to seven dependent global loads per chunk for words that are the same for the whole wave).  Here
//  * a chunk is exactly one block of the packed sequence (64 positions = two code words, one break word, one lower-case word), and
//    everything that is the same for the 64 lanes is scalar: the break and lower-case words are scalar loads, and WHICH lanes hold a
//    valid k-mer / m-mer is mask arithmetic on them in the scalar unit (a break smeared over the k - 1 positions behind it), handed to the
//    lanes as a lane mask -- no per-lane bit extraction, no per-lane position tests;
//  * a lane takes its k-mer out of ONE 16-byte load (the two code words around it; lanes of one half of the wave read the same 16
//    bytes), issued a whole trip ahead; the m-mer that ends at a lane is the low word of its k-mer;
//  * reservations as before (the first lane of a run of equal blocks reserves for the run, two chunks in flight).
// Same buckets as the form above, key for key (tests/test_gpu_parity.py compares the sets; KC_SCATTER_V1=1 runs the old form).
struct KcCover { u64 hi, lo; };   // a 128-bit position mask of two neighbouring blocks: hi = this chunk's 64 positions, lo = the 64 before
__device__ __forceinline__ KcCover kc_cover_shl(KcCover x, int d) {   // d in [0, 63]
    KcCover r;
    r.hi = d ? (x.hi << d) | (x.lo >> (64 - d)) : x.hi;
    r.lo = x.lo << d;
    return r;
}
// OR of x shifted up by 0 .. w - 1 (w in [1, 32]): bit j of the result = some bit of x among the w positions ending at j
__device__ __forceinline__ KcCover kc_cover_smear(KcCover x, int w) {
    KcCover t = x, acc{0, 0};     // t = smear of width 2^j
    int done = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (w & (1 << j)) { const KcCover s = kc_cover_shl(t, done); acc.hi |= s.hi; acc.lo |= s.lo; done += 1 << j; }
        const KcCover u = kc_cover_shl(t, 1 << j);
        t.hi |= u.hi; t.lo |= u.lo;
    }
    return acc;
}
__global__ void __launch_bounds__(256) kc_bucket_scatter_wave2_kernel(const u64 *__restrict__ codes, const u64 *__restrict__ brk, const u64 *__restrict__ low, u64 len, int k,
                                                                      int complements, int case_sensitive, SetView<1> set, u64 *bucket, u64 *bcursor, u64 *counters,
                                                                      int chunks, u64 *sovf = nullptr, u64 sovf_cap = 0, const u64 *__restrict__ boff = nullptr) {
    // boff != nullptr (second pass of the hash-sharded build): the keys of block b land at bucket[boff[b] + position] -- the exact layout,
    // no bucket can run full
    const int lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane((u32)((((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
    const u64 nblk = (len + 63) >> 6;                      // blocks of the packed sequence (bytes behind `len` in the last one are breaks)
    const u64 T0 = (u64)wave * (u64)chunks;                // this wave's first block
    if (T0 >= nblk) return;
    const int m = set.m, win = k - m - 1;
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const u64 maskk = k >= 32 ? ~(u64)0 : (((u64)1 << (2 * k)) - 1);
    const u32 bs = 1u << set.bshift;
    const u32 nblocks = (u32)set.nblocks;
    const int half = lane >> 5, sh = 2 * (31 - (lane & 31));
    const unsigned long long lane_le = ((unsigned long long)2 << lane) - 1;   // lanes at or below this one
    u32 a1p = ~0u, a2p = ~0u, a4p = ~0u, a8p = ~0u;
    bool overflow = false, aside_full = false;
    // the two code words a lane's k-mer lies in (block t: lanes 0..31 -> words 2t-1, 2t; lanes 32..63 -> 2t, 2t+1)
    auto load_codes = [&](u64 t) -> ulonglong2 {
        ulonglong2 v;
        if (t == 0) {   // (wave-uniform: no word in front of the sequence; k-mers of block 0 that are valid lie in the second word of their pair)
            v.x = half ? codes[0] : 0;
            v.y = half ? codes[1] : codes[0];
        } else {
            const u64 *src = codes + 2 * t - 1 + half;
            __builtin_memcpy(&v, src, 16);   // (8-byte aligned)
        }
        return v;
    };
    struct Chunk { u64 key; u32 blk; unsigned long long vm, sm; bool valid, starts; };
    // one chunk: hashes, sliding minimum, this lane's canonical k-mer and block, the wave's run structure.  only_hash: the chunk in
    // front of the wave's first one, for the minima of its first lanes
    auto compute = [&](u64 t, ulonglong2 cw, bool only_hash) -> Chunk {
        Chunk R{0, ~0u, 0, 0, false, false};
        // validity masks (scalar): a position is fine for a k-mer (m-mer) when no break lies among the k (m) positions ending at it
        KcCover B;
        B.hi = brk[t];
        B.lo = t ? brk[t - 1] : ~(u64)0;
        const u64 okm = ~kc_cover_smear(B, m).hi;
        u64 okk = ~kc_cover_smear(B, k).hi;
        if (case_sensitive) {   // the k-mer's FIRST base must be upper case (parser.h:41-42)
            KcCover L;
            L.hi = low[t];
            L.lo = t ? low[t - 1] : 0;
            okk &= ~kc_cover_shl(L, k - 1).hi;
        }
        // the 32 bases ending at this lane, right-aligned
        const u64 v = (cw.y >> sh) | ((cw.x << 1) << (63 - sh));
        const u32 mm = (u32)v & maskm;
        u32 r = __brev(~mm);
        r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
        r >>= 32 - 2 * m;
        const u32 hh = kc_mmer_hash(mm, r);
        const u32 a1 = __builtin_amdgcn_inverse_ballot_w64(okm) ? hh : ~0u;
        // sliding minimum over the win lanes before this one
        u32 tt = kc_halo_get(a1p, a1, lane - 1); const u32 a2 = a1 < tt ? a1 : tt;
        tt = kc_halo_get(a2p, a2, lane - 2); const u32 a4 = a2 < tt ? a2 : tt;
        tt = kc_halo_get(a4p, a4, lane - 4); const u32 a8 = a4 < tt ? a4 : tt;
        u32 pm = ~0u;
        int at = lane - 1;
        if (win & 8) { tt = kc_halo_get(a8p, a8, at); pm = tt < pm ? tt : pm; at -= 8; }
        if (win & 4) { tt = kc_halo_get(a4p, a4, at); pm = tt < pm ? tt : pm; at -= 4; }
        if (win & 2) { tt = kc_halo_get(a2p, a2, at); pm = tt < pm ? tt : pm; at -= 2; }
        if (win & 1) { tt = kc_halo_get(a1p, a1, at); pm = tt < pm ? tt : pm; }
        a1p = a1; a2p = a2; a4p = a4; a8p = a8;
        if (only_hash) return R;
        R.valid = __builtin_amdgcn_inverse_ballot_w64(okk);
        R.vm = okk;
        Key<1> f; f.w[0] = v & maskk;
        R.key = f.w[0];
        if (complements) { const Key<1> rc = key_rc(f, k); if (rc.w[0] < R.key) R.key = rc.w[0]; }
        // block of the minimizer (set_block_home's block part)
        u32 u = pm * 0x85EBCA6Bu; u ^= u >> 16; u *= 0xC2B2AE35u;
        R.blk = R.valid ? __umulhi(u, nblocks) : ~0u;
        const u32 blk_before = (u32)__builtin_amdgcn_update_dpp((int)~0u, (int)R.blk, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        R.starts = R.valid && (lane == 0 || blk_before != R.blk);
        R.sm = __ballot(R.starts);
        return R;
    };
    auto reserve = [&](const Chunk &R) -> u32 {
        if (!R.starts) return 0;
        const unsigned long long stop = (R.sm | ~R.vm) >> lane >> 1;   // the run ends before the next start or invalid lane
        const int rl = stop ? __builtin_ctzll(stop) + 1 : 64 - lane;
        const u64 b = atomicAdd((ull *)&bcursor[R.blk], (ull)rl);
        return b > 0xffffffffULL ? 0xffffffffu : (u32)b;               // (anything from the block size on is an overflow)
    };
    auto commit = [&](const Chunk &R, u32 base) {
        const unsigned long long below = R.sm & lane_le;
        const int leader = below ? 63 - __builtin_clzll(below) : 0;
        base = (u32)__shfl((int)base, leader, 64);
        const u64 at2 = (u64)base + (u64)(lane - leader);
        if (boff) {
            if (R.valid) bucket[boff[R.blk] + at2] = R.key;
            return;
        }
        if (R.valid && at2 < bs) bucket[((u64)R.blk << set.bshift) + at2] = R.key;
        if (kc_scatter_aside<1>(R.valid && at2 >= bs, &R.key, sovf, sovf_cap, counters, aside_full, lane)) overflow = true;
    };
    if (T0) (void)compute(T0 - 1, load_codes(T0 - 1), true);
    u64 tend = T0 + (u64)chunks;
    if (tend > nblk) tend = nblk;
    ulonglong2 ca = load_codes(T0), cb = load_codes(T0 + 1 < tend ? T0 + 1 : T0);
    // two chunks per trip: both reservations are in flight before either is waited for; the next trip's code words are on their way
    for (u64 t = T0; t < tend; t += 2) {
        const bool two = t + 1 < tend;
        const ulonglong2 na = load_codes(t + 2 < tend ? t + 2 : t), nb = load_codes(t + 3 < tend ? t + 3 : t);
        const Chunk A = compute(t, ca, false);
        Chunk Bc{0, ~0u, 0, 0, false, false};
        if (two) Bc = compute(t + 1, cb, false);
        const u32 ba = reserve(A), bb = reserve(Bc);
        commit(A, ba);
        commit(Bc, bb);
        ca = na; cb = nb;
    }
    if (overflow) atomicExch((ull *)&counters[3], 1ULL);
}

// The same for wide keys (k up to 127: the window of the sliding minimum has up to k - 17 = 110 positions, i.e. it reaches two
// chunks back; the rolling form keeps 2 * 110 hashes per thread in LDS there and runs 64 threads per workgroup).
__device__ __forceinline__ bool kc_any_break_long(const u64 *__restrict__ brk, u64 s, int n) {
    bool any = false;
    for (int o = 0; o < n; o += 32) any = any || kc_any_break(brk, s + (u64)o, n - o < 32 ? n - o : 32);
    return any;
}
__device__ __forceinline__ u32 kc_halo_get3(u32 p2, u32 p1, u32 cur, int idx) {   // lane idx of this chunk (idx >= 0), of the previous one (>= -64), of the one before
    const u32 a = __shfl(cur, idx & 63, 64), b = __shfl(p1, idx & 63, 64), c = __shfl(p2, idx & 63, 64);
    return idx >= 0 ? a : (idx >= -64 ? b : c);
}
template <int W>
__global__ void __launch_bounds__(256) kc_bucket_scatter_wave_wide_kernel(const u64 *__restrict__ codes, const u64 *__restrict__ brk, const u64 *__restrict__ low, u64 len, int k,
                                                                          int complements, int case_sensitive, SetView<W> set, u64 *bucket, u64 *bcursor, u64 *counters,
                                                                          int chunks, u64 *sovf = nullptr, u64 sovf_cap = 0) {
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 P0 = wave * (u64)chunks * 64;
    if (P0 >= len) return;
    const int m = set.m, win = k - m - 1;
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const u64 bs = (u64)1 << set.bshift;
    // A[j]: minimum over the 2^j hashes that end at this lane; p1 / p2: the same one / two chunks back
    u32 A[7], p1[7], p2[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) { p1[j] = ~0u; p2[j] = ~0u; }
    bool overflow = false, aside_full = false;
    for (int c = -2; c < chunks; ++c) {
        const long long pp = (long long)P0 + 64LL * c + lane;
        const bool inr = pp >= 0 && (u64)pp < len;
        const u64 p = inr ? (u64)pp : 0;
        if (c >= 0 && P0 + 64ULL * (u64)c >= len) break;   // (wave-uniform)
        u32 h = ~0u;
        if (inr && p + 1 >= (u64)m && !kc_any_break(brk, p + 1 - (u64)m, m)) {
            const u32 mm = (u32)kc_extract_bases(codes, p + 1 - (u64)m, m);
            u32 r = __brev(~mm);
            r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
            r >>= 32 - 2 * m;
            h = kc_mmer_hash(mm & maskm, r);
        }
        bool valid = false;
        Key<W> key = key_zero<W>();
        if (c >= 0 && inr && p + 1 >= (u64)k && !kc_any_break_long(brk, p + 1 - (u64)k, k)) {
            const u64 fp = p + 1 - (u64)k;
            valid = !case_sensitive || !((low[fp >> 6] >> (fp & 63)) & 1);
            if (valid) {
#pragma unroll
                for (int j = 0; j < W; ++j) {   // word j: the bases 32 j ... 32 j + 31 counted from the END of the k-mer
                    const int nb = k - 32 * j >= 32 ? 32 : k - 32 * j;
                    key.w[j] = nb > 0 ? kc_extract_bases(codes, p + 1 - (u64)(32 * j) - (u64)nb, nb) : 0;
                }
                if (complements) { const Key<W> r = key_rc(key, k); if (key_lt(r, key)) key = r; }
            }
        }
        A[0] = h;
#pragma unroll
        for (int j = 1; j < 7; ++j) {
            const int d = 1 << (j - 1);
            const u32 a = __shfl(A[j - 1], (lane - d) & 63, 64), b = __shfl(p1[j - 1], (lane - d) & 63, 64);
            const u32 t = lane - d >= 0 ? a : b;
            A[j] = A[j - 1] < t ? A[j - 1] : t;
        }
        u32 pm = ~0u;
        int at = lane - 1;
#pragma unroll
        for (int j = 6; j >= 0; --j)
            if (win & (1 << j)) { const u32 t = kc_halo_get3(p2[j], p1[j], A[j], at); pm = t < pm ? t : pm; at -= 1 << j; }
#pragma unroll
        for (int j = 0; j < 7; ++j) { p2[j] = p1[j]; p1[j] = A[j]; }
        if (c < 0) continue;
        const u64 blk = valid ? set_block_home(set, pm, 0) >> set.bshift : ~(u64)0;
        const u64 blk_before = __shfl_up(blk, 1, 64);
        const bool starts = valid && (lane == 0 || blk_before != blk);
        const unsigned long long vm = __ballot(valid), sm = __ballot(starts);
        if (!vm) continue;
        u64 base = 0;
        if (starts) {
            const unsigned long long stop = (sm | ~vm) >> lane >> 1;
            const int rl = stop ? __builtin_ctzll(stop) + 1 : 64 - lane;
            base = atomicAdd((ull *)&bcursor[blk], (ull)rl);
        }
        const unsigned long long below = sm & (((unsigned long long)2 << lane) - 1);
        const int leader = below ? 63 - __builtin_clzll(below) : 0;
        base = __shfl(base, leader, 64);
        const u64 at2 = base + (u64)(lane - leader);
        if (valid && at2 < bs) {
            u64 *dst = bucket + ((blk << set.bshift) + at2) * W;
#pragma unroll
            for (int w = 0; w < W; ++w) dst[w] = key.w[w];
        }
        if (kc_scatter_aside<W>(valid && at2 >= bs, key.w, sovf, sovf_cap, counters, aside_full, lane)) overflow = true;
    }
    if (overflow) atomicExch((ull *)&counters[3], 1ULL);
}

// one workgroup per table block; dynamic LDS: (1 << bshift) keys of W words + (1 << bshift) tag bytes.  A slot is claimed
// by an LDS compare-and-swap on its MOST significant key word (never all ones: 2k < 64 W), then the other words and
// finally the tag byte are written; a lane that finds "its" top word in an unpublished slot looks again in its next loop
// iteration (nobody waits inside an iteration, so lanes of one wave cannot block each other).
#define KC_MAX_SEG 64
#define KC_MAX_DYN_LDS (160 * 1024 - 4096)   /* dynamic LDS limit requested for the build kernels (their static LDS is ~1 KB) */
template <int W, bool IDS = false>
__global__ void __launch_bounds__(1024, W == 1 ? 8 : 4) kc_block_build_kernel(SetView<W> set, const u64 *__restrict__ bucket, const u64 *__restrict__ bcursor,
                                                              int fresh, u64 *counters, u64 *ovf_list, u64 ovf_cap,
                                                              const u64 *__restrict__ seg_scan, int nseg, int merge_r, int sample_stride,
                                                              u32 *ncount_out = nullptr) {
    // IDS: the block leaves with its dense node ids prepared (kc_set_dev.h): key word 0 of every empty slot = the
    // number of occupied slots before it, ncount_out[block] = its k-mers, counters[4] = the largest block.
    // nseg == 0: the block's keys are bucket[block * 2^bshift ...], bcursor[block] of them (single GPU).  nseg > 0 (hash-sharded
    // build): the block is LOCAL block blockIdx.x of this rank's range and its keys arrived in nseg segments, one per source
    // rank: segment q = bucket[seg_scan[q * gridDim.x + block] .. seg_scan[q * gridDim.x + block + 1]).
    // merge_r > 0 (the scatter ran with merge_r times as many blocks as the table now has -- the table was sized by k-mer
    // OCCURRENCES and is now sized by distinct k-mers): block b is built from the merge_r scatter buckets b * merge_r ... (the
    // block index is a range reduction of the minimizer hash, so dividing the number of blocks by r maps bucket i to block i / r).
    // sample_stride > 0: count only -- workgroup i inserts scatter bucket i * sample_stride into LDS and adds the number of
    // distinct keys to counters[2] and the bucket's occurrences to counters[3]; nothing is written to the table.
    // Round 5 (PMC counters of this kernel at 10^9 keys): it is bound by instruction ISSUE -- 986 vector and 1158 scalar instructions per
    // wave, the scalar ones mostly exec-mask bookkeeping of the divergent probe loop; LDS 32 % busy, HBM at 2.3 TB/s.  Persistent
    // workgroups that fetch the next block's keys under the write-out of the current one were built and measured: slower (26.6 against
    // 23.5 ms at 2.5 * 10^9), so a block is a workgroup.
    extern __shared__ u64 kc_blk[];
    __shared__ u64 seg_lo[KC_MAX_SEG], seg_pre[KC_MAX_SEG + 1];
    const u32 bs = 1u << set.bshift;
    u64 *lk = kc_blk;
    // (wide keys publish a slot through its tag byte while other lanes spin on it: volatile accesses.  The compiler turns those into FLAT
    // instructions with system-scope bits, each followed by a wait for ALL outstanding memory operations of the wave -- in the write-out
    // that serialised a wave's rows behind its own global stores.  Single-word keys need no such protocol: `ltp` is the plain LDS pointer,
    // every access of theirs goes through it; round 5: 28.8 -> see DESIGN.md)
    volatile u8 *lt = reinterpret_cast<volatile u8 *>(kc_blk + (size_t)bs * W);
    u8 *ltp = reinterpret_cast<u8 *>(kc_blk + (size_t)bs * W);
    __shared__ u32 added_s;
    if (merge_r) nseg = merge_r;
    // single-word keys of a plain bucket: all keys of a thread (8 x 1024 = a whole bucket of 2^13 keys; more are loaded in the loop)
    // are on their way while the LDS table is set up
    const bool pf = W == 1 && nseg == 0 && !(fresh & 2);   // (bit 1 of `fresh`: test knob, no prefetch)
    const bool st16 = !(fresh & 4);                          // (bit 2: test knob, 8-byte stores in the write-out)
    const int exp_ = fresh >> 4;                             // (timing experiments, KC_BUILD_EXP: 1 no inserts, 2 no write-out, 4 no key loads -- the results are wrong then)
    fresh &= 1;
    constexpr int NPRE = 8;
    u64 pre[NPRE];
#pragma unroll
    for (int t = 0; t < NPRE; ++t) pre[t] = 0;
    const u32 nblk = gridDim.x, vb = blockIdx.x;
    const u64 src_blk = sample_stride ? (u64)vb * (u64)sample_stride : (u64)vb;
    const u64 base = set.rlo + (src_blk << set.bshift);
    u64 n_pre = 0;                                           // keys of the block (pf)
    auto prefetch = [&](u64 n0) {
        const u64 *bk = bucket + base;
        // (no branches: a conditional load is a basic block of its own, and the compiler put a wait for ALL outstanding memory operations
        // in front of each -- eight round trips in a row; the threads beyond the count re-read key 0)
#pragma unroll
        for (int t = 0; t < NPRE; ++t) {
            const u32 j = threadIdx.x + 1024u * (u32)t;
            pre[t] = bk[j < n0 ? j : 0u];
        }
    };
    if (pf) {
        n_pre = bcursor[src_blk];
        if (n_pre > bs) n_pre = bs;
        if (!(exp_ & 4)) prefetch(n_pre);
    }
    if (threadIdx.x == 0) {
        added_s = 0;
        u64 a = 0;
        for (int q = 0; q < nseg; ++q) {
            u64 lo, hi;
            if (merge_r) {
                const u64 sb = (u64)vb * (u64)merge_r + (u64)q;
                u64 c = bcursor[sb];
                if (c > bs) c = bs;
                lo = sb << set.bshift; hi = lo + c;
            } else { lo = seg_scan[(u64)q * nblk + vb]; hi = seg_scan[(u64)q * nblk + vb + 1]; }
            seg_lo[q] = lo; seg_pre[q] = a; a += hi - lo;
        }
        seg_pre[nseg] = a;
    }
    for (u32 i = threadIdx.x; i < bs / 8; i += blockDim.x)
        reinterpret_cast<u64 *>(ltp)[i] = fresh ? 0 : reinterpret_cast<const u64 *>(set.tags + base)[i];
    if (!fresh) __syncthreads();   // (a fresh block's keys do not look at the tags)
    for (u32 i = threadIdx.x; i < bs * W; i += blockDim.x) {
        // keys of empty slots are undefined in the wide-key table (their state says so): only the claim word matters here
        const bool occupied = !fresh && ltp[i / W] != 0;
        lk[i] = occupied ? set.keys[base * W + i] : ~(u64)0;
    }
    __syncthreads();
    u64 n;
    if (nseg) n = seg_pre[nseg];
    else if (pf) n = n_pre;
    else { n = bcursor[src_blk]; if (n > bs) n = bs; }   // (the scatter kernel has flagged the overflow; the caller re-inserts everything through the global path)
    if (exp_ & 1) n = 0;
    u32 added = 0;
    auto insert_key = [&](const Key<W> &key) {
        const u64 top = key.w[W - 1];
        u32 off; u8 tag;
        set_place_local(set, key, &off, &tag);
        bool done = false;
        u32 p = 0;
        if constexpr (W == 1) {
            // Straight-line probe rounds (round 5): the kernel is bound by instruction issue, not by LDS or HBM (PMC: 986 vector + 1158
            // scalar instructions per wave, the scalar ones mostly exec-mask bookkeeping of the divergent loop below; LDS 32 % busy).
            // Every lane of the wave issues the compare-and-swap in every round: a lane that is through keeps issuing it at its last slot,
            // which holds a key by then -- its own, its duplicate, or the key in the way where it gave up -- so the swap fails and changes
            // nothing (no per-lane selects, no exec-mask bookkeeping); the wave leaves when no lane moves on, and the tag is written once,
            // behind the rounds.
            u32 sl = off;
            const u32 lim = off + set.local_probes;
            bool took = false, gave_up = false, adv;
            do {
                const u64 cur = atomicCAS((ull *)&lk[sl], ~0ULL, (ull)top);
                took = took || cur == ~(u64)0;
                const bool miss = cur != ~(u64)0 && cur != top;   // another key lives here
                adv = miss && sl + 1 < lim;
                gave_up = gave_up || (miss && !adv);
                sl += adv ? 1u : 0u;
            } while (__ballot(adv));
            if (took) { ltp[sl] = tag; ++added; }
            // (rounds that look at the tags of 8 slots and go straight for the first empty slot or the first with the key's own tag -- ~1.3
            // rounds per key instead of ~8 -- were built and measured: 23.4 against 23.3 ms.  KC_BUILD_EXP: of the 23.3 ms, 2.0 set the LDS
            // table up, 1.9-3.5 load the keys, 3.9 write the block out and 11.5 insert -- and most of that is the key's place hash.)
            done = !gave_up;
        } else
        while (p < set.local_probes && !done) {
            const u32 s = off + p;
            u64 cur = lk[(size_t)s * W + W - 1];
            if (cur == ~(u64)0) cur = atomicCAS((ull *)&lk[(size_t)s * W + W - 1], ~0ULL, (ull)top);
            if (cur == ~(u64)0) {   // claimed: publish
                if constexpr (W == 1) ltp[s] = tag;   // (nobody reads a tag before the barrier behind the loop)
                else {
                    for (int w = 0; w < W - 1; ++w) lk[(size_t)s * W + w] = key.w[w];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __hip_atomic_store(&ltp[s], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (an LDS store; the volatile one was a FLAT instruction behind a full wait)
                }
                ++added;
                done = true;
            } else if (cur == top) {
                if constexpr (W == 1) done = true;
                else {
                    if (__hip_atomic_load(&ltp[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) continue;   // being published by another lane: look again (p unchanged)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    bool eq = true;
                    for (int w = 0; w < W - 1; ++w) eq = eq && (lk[(size_t)s * W + w] == key.w[w]);
                    if (eq) done = true; else ++p;
                }
            } else ++p;
        }
        if (!done && !sample_stride) {   // block-local run exhausted: the global insert (second home) takes it after all blocks are written
            const u64 pos = atomicAdd((ull *)&counters[2], 1ULL);
            if (pos < ovf_cap)
                for (int w = 0; w < W; ++w) ovf_list[pos * W + w] = key.w[w];
        }
    };
    u64 jrest = threadIdx.x;
    if (pf) {
        // the prefetched keys by their register (a select chain over pre[] per trip was 14 instructions of every insert)
#pragma unroll
        for (int t = 0; t < NPRE; ++t)
            if ((u64)threadIdx.x + 1024u * (u64)t < n) { Key<W> key; key.w[0] = pre[t]; insert_key(key); }
        jrest += 1024u * (u64)NPRE;
    }
    for (u64 j = jrest; j < n; j += blockDim.x) {
        Key<W> key;
        u64 src = base + j;
        if (nseg) {
            int q = 0;
            while (q + 1 < nseg && j >= seg_pre[q + 1]) ++q;
            src = seg_lo[q] + (j - seg_pre[q]);
        }
        for (int w = 0; w < W; ++w) key.w[w] = bucket[src * W + w];
        insert_key(key);
    }
    if (added) atomicAdd(&added_s, added);
    __syncthreads();
    if (sample_stride) {
        if (threadIdx.x == 0) { atomicAdd((ull *)&counters[2], (ull)added_s); atomicAdd((ull *)&counters[3], (ull)n); }
        return;
    }
    if (exp_ & 2) return;
    if (IDS) {
        // Write-out with the dense node ids prepared on the way: wave w owns the slots [w bs/16, (w+1) bs/16) and writes them in
        // rows of 64 (512 contiguous bytes per row of single-word keys); an empty slot leaves with the number of occupied slots
        // before it (ballot + popcount on top of the count of the waves before) instead of the empty marker.
        const u32 nwv = blockDim.x >> 6, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const u32 per_wave = bs / nwv, rows = (per_wave + 63) / 64;   // (per_wave is a multiple of 8: blocks have >= 512 slots)
        const unsigned long long ltm = ((unsigned long long)1 << lane) - 1;
        // occupied slots in the shares of the waves before this one: every wave counts them for itself, 8 tags per LDS load (no
        // barrier, no waiting for one another -- the block's other workgroup on the CU is in its insert phase meanwhile)
        u32 at = 0;
        {
            const u64 *lt64 = reinterpret_cast<const u64 *>(ltp);
            const u32 words_before = wv * per_wave / 8;
            for (u32 i = lane; i < words_before; i += 64) at += (u32)__popcll(lt64[i] & 0x8080808080808080ULL);   // (tag bytes of occupied slots have bit 7 set)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) at += __shfl_xor(at, d, 64);
        }
        if (W == 1 && st16 && per_wave % 128 == 0) {
            // two neighbouring slots per lane: 16-byte LDS loads and 16-byte stores (1 KB per wave instruction)
            for (u32 r = 0; r < per_wave / 128; ++r) {
                const u32 sl = wv * per_wave + r * 128 + 2 * lane;
                const u32 tg = *reinterpret_cast<const unsigned short *>(ltp + sl);
                const bool occ0 = (tg & 0xffu) != 0, occ1 = (tg >> 8) != 0;
                const unsigned long long mk0 = __ballot(occ0), mk1 = __ballot(occ1);
                const u64 cnt0 = (u64)(at + (u32)__popcll(mk0 & ltm) + (u32)__popcll(mk1 & ltm)), cnt1 = cnt0 + (occ0 ? 1 : 0);
                ulonglong2 v;
                v.x = lk[sl]; v.y = lk[sl + 1];
                if (!occ0) v.x = KC_ANN_MARK | cnt0;
                if (!occ1) v.y = KC_ANN_MARK | cnt1;
                *reinterpret_cast<ulonglong2 *>(set.keys + base + sl) = v;
                at += (u32)__popcll(mk0) + (u32)__popcll(mk1);
            }
        } else
        for (u32 r = 0; r < rows; ++r) {
            const u32 off = r * 64 + lane, sl = wv * per_wave + off;
            const bool in = off < per_wave, occ = in && ltp[sl] != 0;
            const unsigned long long mk = __ballot(occ);
            if (in) {
                const u64 cnt = (u64)(at + (u32)__popcll(mk & ltm));
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    u64 v = lk[(size_t)sl * W + w];
                    if (!occ && w == 0) v = W == 1 ? (KC_ANN_MARK | cnt) : cnt;
                    set.keys[(base + sl) * W + w] = v;
                }
                if (W > 1) set.state[base + sl] = occ ? KC_ST_FULL : KC_ST_EMPTY;
            }
            at += (u32)__popcll(mk);
        }
        for (u32 i = threadIdx.x; i < bs / 8; i += blockDim.x)
            reinterpret_cast<u64 *>(set.tags + base)[i] = reinterpret_cast<const u64 *>(ltp)[i];
        if (threadIdx.x == blockDim.x - 64) ncount_out[vb] = at;   // (the last wave ends on the block's total; the largest block: kc_max_u32_kernel)
    } else {
        for (u32 i = threadIdx.x; i < bs * W; i += blockDim.x) set.keys[base * W + i] = lk[i];
        for (u32 i = threadIdx.x; i < bs / 8; i += blockDim.x)
            reinterpret_cast<u64 *>(set.tags + base)[i] = reinterpret_cast<const u64 *>(ltp)[i];
        if (W > 1)
            for (u32 i = threadIdx.x; i < bs; i += blockDim.x) set.state[base + i] = ltp[i] ? KC_ST_FULL : KC_ST_EMPTY;
    }
    if (vb == 0 && threadIdx.x < 8) set.tags[set.rlo + set.rcap + threadIdx.x] = ltp[threadIdx.x];   // mirror for window loads that wrap at the end of the range
    if (threadIdx.x == 0 && added_s) atomicAdd((ull *)&counters[0], (ull)added_s);
}

// hash-sharded -z (ReadKMersFiltered / AddKMersWithFrequencies, parser.h:51-85,122-141): as kc_block_build_kernel for a block whose keys
// arrived in nseg segments, but only the keys that occur at least min_count times among them stay.  Pass 1 inserts every key and
// counts its occurrences (a 16-bit counter per slot behind the tags, two to a word; counting stops at 256: min_count <= 255) and
// notes for every received key the slot it ended up in (`where`, 2 B per key); the table is cleared; pass 2 inserts the keys whose
// slot counted min_count or more.  A key that finds no room in its block's probe run raises counters[3]: the spill list has no
// counters, the caller leaves the sharded path.
template <int W>
__global__ void __launch_bounds__(1024) kc_block_build_filter_kernel(SetView<W> set, const u64 *__restrict__ bucket, u64 *counters, const u64 *__restrict__ seg_scan,
                                                                     int nseg, u32 min_count, unsigned short *where) {
    extern __shared__ u64 kc_blk[];
    __shared__ u64 seg_lo[KC_MAX_SEG], seg_pre[KC_MAX_SEG + 1];
    __shared__ u32 added_s;
    const u32 bs = 1u << set.bshift;
    u64 *lk = kc_blk;
    volatile u8 *lt = reinterpret_cast<volatile u8 *>(kc_blk + (size_t)bs * W);
    u8 *ltp = reinterpret_cast<u8 *>(kc_blk + (size_t)bs * W);
    u32 *cnt = reinterpret_cast<u32 *>(ltp + bs);   // bs / 2 words
    const u64 base = set.rlo + ((u64)blockIdx.x << set.bshift);
    if (threadIdx.x == 0) {
        added_s = 0;
        u64 a = 0;
        for (int q = 0; q < nseg; ++q) {
            const u64 lo = seg_scan[(u64)q * gridDim.x + blockIdx.x], hi = seg_scan[(u64)q * gridDim.x + blockIdx.x + 1];
            seg_lo[q] = lo; seg_pre[q] = a; a += hi - lo;
        }
        seg_pre[nseg] = a;
    }
    auto clear = [&]() {
        for (u32 i = threadIdx.x; i < bs / 8; i += blockDim.x) reinterpret_cast<u64 *>(ltp)[i] = 0;
        for (u32 i = threadIdx.x; i < bs * W; i += blockDim.x) lk[i] = ~(u64)0;
    };
    clear();
    for (u32 i = threadIdx.x; i < bs / 2; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const u64 n = seg_pre[nseg];
    auto src_of = [&](u64 j) -> u64 {
        int q = 0;
        while (q + 1 < nseg && j >= seg_pre[q + 1]) ++q;
        return seg_lo[q] + (j - seg_pre[q]);
    };
    // insert `key` (or find it): the slot, or 0xffff when the block-local probe run is exhausted; *claimed = this call put it there
    auto insert = [&](const Key<W> &key, bool *claimed) -> u32 {
        const u64 top = key.w[W - 1];
        u32 off; u8 tag;
        set_place_local(set, key, &off, &tag);
        *claimed = false;
        if constexpr (W == 1) {   // straight-line probe rounds, as in kc_block_build_kernel (every lane of the wave swaps every round)
            u32 sl = off;
            const u32 lim = off + set.local_probes;
            bool took = false, gave_up = false, adv;
            do {
                const u64 cur = atomicCAS((ull *)&lk[sl], ~0ULL, (ull)top);
                took = took || cur == ~(u64)0;
                const bool miss = cur != ~(u64)0 && cur != top;   // another key lives here
                adv = miss && sl + 1 < lim;
                gave_up = gave_up || (miss && !adv);
                sl += adv ? 1u : 0u;
            } while (__ballot(adv));
            if (took) ltp[sl] = tag;
            *claimed = took;
            return gave_up ? 0xffffu : sl;
        }
        u32 p = 0;
        while (p < set.local_probes) {
            const u32 s = off + p;
            u64 cur = lk[(size_t)s * W + W - 1];
            if (cur == ~(u64)0) cur = atomicCAS((ull *)&lk[(size_t)s * W + W - 1], ~0ULL, (ull)top);
            if (cur == ~(u64)0) {   // claimed: publish
                if constexpr (W == 1) ltp[s] = tag;   // (single-word keys: nobody waits on a tag; plain LDS store, see kc_block_build_kernel)
                else {
                    for (int w = 0; w < W - 1; ++w) lk[(size_t)s * W + w] = key.w[w];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    lt[s] = tag;
                }
                *claimed = true;
                return s;
            } else if (cur == top) {
                if constexpr (W == 1) return s;
                if (W > 1 && lt[s] == 0) continue;   // being published by another lane: look again (p unchanged)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                bool eq = true;
                for (int w = 0; w < W - 1; ++w) eq = eq && (lk[(size_t)s * W + w] == key.w[w]);
                if (eq) return s;
                ++p;
            } else ++p;
        }
        return 0xffffu;
    };
    bool lost = false;
    for (u64 j = threadIdx.x; j < n; j += blockDim.x) {
        const u64 src = src_of(j);
        Key<W> key;
        for (int w = 0; w < W; ++w) key.w[w] = bucket[src * W + w];
        bool claimed;
        const u32 s = insert(key, &claimed);
        where[src] = (unsigned short)s;
        if (s == 0xffffu) { lost = true; continue; }
        u32 *c = &cnt[s >> 1];
        const u32 sh = 16u * (s & 1u);
        if (((__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xffffu) < 256u) atomicAdd(c, 1u << sh);
    }
    if (lost) atomicExch((ull *)&counters[3], 1ULL);
    __syncthreads();
    clear();
    __syncthreads();
    u32 added = 0;
    for (u64 j = threadIdx.x; j < n; j += blockDim.x) {
        const u64 src = src_of(j);
        const u32 s0 = where[src];
        if (s0 == 0xffffu || ((cnt[s0 >> 1] >> (16u * (s0 & 1u))) & 0xffffu) < min_count) continue;
        Key<W> key;
        for (int w = 0; w < W; ++w) key.w[w] = bucket[src * W + w];
        bool claimed;
        // (the survivors are a subset of what fitted in pass 1, but bounded probe runs depend on the insertion order: a survivor that
        // finds no room now is reported like one in pass 1 -- the ranks then leave the sharded path together)
        if (insert(key, &claimed) == 0xffffu) lost = true;
        if (claimed) ++added;
    }
    if (added) atomicAdd(&added_s, added);
    if (lost) atomicExch((ull *)&counters[3], 1ULL);
    __syncthreads();
    for (u32 i = threadIdx.x; i < bs * W; i += blockDim.x) set.keys[base * W + i] = lk[i];
    for (u32 i = threadIdx.x; i < bs / 8; i += blockDim.x)
        reinterpret_cast<u64 *>(set.tags + base)[i] = reinterpret_cast<const u64 *>(ltp)[i];
    if (W > 1)
        for (u32 i = threadIdx.x; i < bs; i += blockDim.x) set.state[base + i] = ltp[i] ? KC_ST_FULL : KC_ST_EMPTY;
    if (blockIdx.x == 0 && threadIdx.x < 8) set.tags[set.rlo + set.rcap + threadIdx.x] = ltp[threadIdx.x];   // mirror for window loads that wrap at the end of the range
    if (threadIdx.x == 0 && added_s) atomicAdd((ull *)&counters[0], (ull)added_s);
}

template <int W>
__global__ void __launch_bounds__(256) kc_insert_list_kernel(SetView<W> set, const u64 *__restrict__ list, u64 n, u64 *counters, u64 *blocks_out = nullptr) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Key<W> key;
    for (int w = 0; w < W; ++w) key.w[w] = list[t * W + w];
    u64 slot = 0;
    int r = set_insert(set, key, &slot);
    if (blocks_out) blocks_out[t] = r >= 0 ? (slot - set.rlo) >> set.bshift : 0;   // (the block it went to: its node numbering is redone, kc_annotate_kernel)
    if (r > 0) atomicAdd((ull *)&counters[0], 1ULL);
    else if (r < 0) atomicExch((ull *)&counters[1], 1ULL);
}

// re-insert into a new table; with counts: carry the occurrence counters, and keep only k-mers seen >= min_count times
template <int W>
__global__ void __launch_bounds__(256) kc_rehash_kernel(SetView<W> from, SetView<W> to, u64 *counters, const u32 *from_counts,
                                                        u32 *to_counts, u32 min_count) {
    u64 t = kc_gtid();
    if (t >= from.cap) return;
    if (!set_slot_full(from, t)) return;
    if (from_counts && from_counts[t] < min_count) return;
    Key<W> key = set_key_at(from, t);
    u64 slot;
    int r = set_insert(to, key, &slot);
    if (r < 0) atomicExch((ull *)&counters[1], 1ULL);
    else if (to_counts) to_counts[slot] = from_counts[t];
    if (r > 0 && min_count > 0) atomicAdd((ull *)&counters[2], 1ULL);   // survivors of a filter pass
}

template <int W>
__global__ void __launch_bounds__(256) kc_export_kernel(SetView<W> set, u64 *out, u64 *cursor) {
    u64 t = kc_gtid();
    if (t >= set.cap) return;
    if (!set_slot_full(set, t)) return;
    u64 pos = atomicAdd((ull *)cursor, 1ULL);
    for (int i = 0; i < W; ++i) out[pos * W + i] = set.keys[t * W + i];
}

template <int W>
__global__ void __launch_bounds__(256) kc_contains_kernel(SetView<W> set, const u64 *keys, u64 n, u8 *out) {
    u64 t = kc_gtid();
    if (t >= n) return;
    Key<W> key;
    for (int i = 0; i < W; ++i) key.w[i] = keys[t * W + i];
    out[t] = set_find(set, key) != KC_NONE64 ? 1 : 0;
}

static int set_fill_table(kc_ctx *ctx, int W, u64 cap, u64 *keys, u32 *state, u8 *tags) {
    KC_HIP(ctx, hipMemsetAsync(tags, 0, cap + 8, ctx->stream));
    if (W == 1) { KC_HIP(ctx, hipMemsetAsync(keys, 0xff, cap * 8, ctx->stream)); }
    else { KC_HIP(ctx, hipMemsetAsync(state, 0, cap * sizeof(u32), ctx->stream)); }
    return KC_OK;
}
static int set_alloc_table(kc_set *set, u64 cap, u64 **keys, u32 **state, u8 **tags, bool fill = true) {
    kc_ctx *ctx = set->ctx;
    KC_TRY(kc_alloc(ctx, keys, cap * set->W));
    KC_TRY(kc_alloc(ctx, tags, cap + 8));
    *state = nullptr;
    if (set->W != 1) KC_TRY(kc_alloc(ctx, state, cap));
    if (fill) KC_TRY(set_fill_table(ctx, set->W, cap, *keys, *state, *tags));
    return KC_OK;
}
int kcs_clean_table(const kc_set *s) {
    if (!s->unclean) return KC_OK;
    s->unclean = false;
    return set_fill_table(s->ctx, s->W, s->rcap ? s->rcap : s->cap, s->keys, s->state, s->tags);
}

static int bucketed_max_bshift(int W) { return W == 1 ? 14 : (W == 2 ? 13 : 12); }

extern "C" int kc_set_create(kc_ctx *ctx, int k, int complements, uint64_t capacity_hint, kc_set **out) {
    if (!ctx || !out) return KC_E_ARG;
    *out = nullptr;
    if (k < 1 || k > 127) return kc_fail(ctx, KC_E_ARG, "k must be in 1..127");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    kc_set *s = new kc_set();
    s->ctx = ctx; s->k = k; s->W = kc_words_for_k(k); s->complements = complements ? 1 : 0;
    s->mlen = kc_default_mlen(k);
    // slots per block: as large as the LDS of a CU allows for the bucketed build (128 KB of keys + the tags): 16384 / 8192 /
    // 4096 for 1 / 2 / 4 key words; large blocks keep the skewed minimizer popularity from overflowing them
    s->bshift = bucketed_max_bshift(s->W);
    if (s->W == 1) s->bshift = 13;   // 8192-slot blocks (72 KB of LDS): two build workgroups per CU overlap their load / insert / store
                                    // phases; measured +5 % end to end at 2.5 G k-mers, no difference at 100 M
    if (const char *e = getenv("KC_MLEN")) { int v = atoi(e); s->mlen = (v > 0 && v <= 16 && v + 2 <= k) ? v : 0; }   // tuning/ablation knobs
    if (const char *e = getenv("KC_BSHIFT")) { int v = atoi(e); if (v >= 9 && v <= 15) s->bshift = v; }   // (the probe kernel stages a block's tags in LDS)
    if (const char *e = getenv("KC_LOCAL_PROBES")) { int v = atoi(e); if (v >= 1) s->local_probes = (u32)v; }
    // target load 0.5 (shorter tag scans, measured 10 % faster neighbour probes) when table + links at that load
    // stay below 60 % of the device memory, else 0.66
    double load = 0.5;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
        // per slot: keys + tag (+ state word for wide keys) + the larger of the build buckets (8W, freed before phase 1) and
        // the port links (16)
        double per_slot = 8.0 * s->W + 1 + (s->W > 1 ? 4 : 0) + (8.0 * s->W > 16 ? 8.0 * s->W : 16);
        double need = (double)capacity_hint / load * per_slot + (double)capacity_hint * 4;
        if (total_b && need > 0.6 * (double)total_b) load = 0.66;
    }
    if (const char *e = getenv("KC_LOAD")) { double v = atof(e); if (v >= 0.2 && v <= 0.9) load = v; }
    u64 cap = (u64)((double)capacity_hint / load) + 1024;
    s->nblocks = (cap + (((u64)1 << s->bshift) - 1)) >> s->bshift;
    if (s->nblocks >= 4096) s->nblocks = (s->nblocks + 63) / 64 * 64;   // (a table sized by occurrences can later be rebuilt with 2, 4, ... 64 times fewer blocks)
    cap = s->nblocks << s->bshift;
    s->cap = cap;
    int rc = set_alloc_table(s, cap, &s->keys, &s->state, &s->tags, false);   // filled on first use (kc_view), or never (bucketed build)
    s->unclean = true;
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->counters, 8);   // [0] size, [1] capacity flag, [2] [3] scratch of the builds, [4] largest block (dense ids)
    if (rc == KC_OK && hipMemsetAsync(s->counters, 0, 8 * sizeof(u64), ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
    if (rc != KC_OK) { kc_set_destroy(s); return rc; }
    *out = s;
    return KC_OK;
}
extern "C" void kc_set_destroy(kc_set *set) {
    if (!set) return;
    kc_ctx *ctx = set->ctx;
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, set->keys);
    kc_release(ctx, set->state);
    kc_release(ctx, set->tags);
    kc_release(ctx, set->counts);
    kc_release(ctx, set->counters);
    kc_release(ctx, set->ncount);
    delete set;
}

// ---- dense node ids (phase 1 numbers the k-mers of a block in slot order; kc_set_dev.h, kc_simp_dev.h).  One workgroup per block:
// ncount[b] = occupied slots of block b (its maximum into counters[4]), and key word 0 of every empty slot = the number of occupied
// slots before it in the block (single-word keys: under KC_ANN_MARK).  The block-build kernel of the bucketed path leaves the same
// words behind on its way (dense_ok stays set); this sweep serves tables that saw a direct insert since.
static void kcs_void_dense(kc_set *set) {
    set->dense_ok = false;
    set->homes_primary = false;
    if (set->ncount) { (void)hipStreamSynchronize(set->ctx->stream); kc_release(set->ctx, set->ncount); set->ncount = nullptr; }
}
template <int W>
__global__ void __launch_bounds__(256) kc_annotate_kernel(SetView<W> set, u32 *ncount, u64 *max_out, const u64 *__restrict__ only = nullptr, u32 *claim = nullptr) {
    // only != nullptr: the listed blocks (the ones that took spilled keys after the build kernel had numbered them); a block listed
    // several times is numbered by the first workgroup that claims it (claim[]: one zero-initialised word per block)
    const u32 bs = 1u << set.bshift, per = bs >> 8;   // (blocks have >= 512 slots)
    const u64 blk = only ? only[blockIdx.x] : (u64)blockIdx.x;
    if (claim) {
        __shared__ u32 taken;
        if (threadIdx.x == 0) taken = atomicExch(&claim[blk], 1u);
        __syncthreads();
        if (taken) return;
    }
    const u64 base = set.rlo + (blk << set.bshift);
    const u8 *tg = set.tags + base + (u64)threadIdx.x * per;
    u32 c = 0;
    for (u32 i = 0; i < per; ++i) c += tg[i] != 0 ? 1u : 0u;
    u64 tot;
    u64 at = kc_block_excl_scan((u64)c, &tot);
    for (u32 i = 0; i < per; ++i) {
        if (tg[i] != 0) { ++at; continue; }
        set.keys[(base + (u64)threadIdx.x * per + i) * W] = W == 1 ? (KC_ANN_MARK | at) : at;
    }
    if (threadIdx.x == 0) { ncount[blk] = (u32)tot; if (tot > *(volatile u64 *)max_out) atomicMax((ull *)max_out, (ull)tot); }
}
__global__ void __launch_bounds__(256) kc_max_u32_kernel(const u32 *__restrict__ a, u64 n, u64 *out) {
    u32 m = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) m = a[i] > m ? a[i] : m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const u32 o = __shfl_xor(m, d, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax((ull *)out, (ull)m);
}
static u32 kcs_stride_for(u64 max_count) { return (u32)((max_count + 7) / 8 * 8 + (max_count == 0 ? 8 : 0)); }   // (a multiple of 8: every block's link words start on a 128-byte line)
int kcs_prepare_dense(kc_set *set) {
    kc_ctx *ctx = set->ctx;
    const u64 nblk = (set->rcap ? set->rcap : set->cap) >> set->bshift;
    if (set->dense_ok && set->ncount) return KC_OK;
    if (set->bshift < 9) return kc_fail(ctx, KC_E_ARG, "blocks of fewer than 512 slots");
    if (set->ncount) { kc_release(ctx, set->ncount); set->ncount = nullptr; }
    KC_TRY(kc_alloc(ctx, &set->ncount, nblk + 2));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 4, 0, 8, ctx->stream));
    switch (set->W) {
        case 1: KC_LAUNCH(kc_annotate_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, kc_view<1>(set), set->ncount, set->counters + 4); break;
        case 2: KC_LAUNCH(kc_annotate_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, kc_view<2>(set), set->ncount, set->counters + 4); break;
        default: KC_LAUNCH(kc_annotate_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, kc_view<4>(set), set->ncount, set->counters + 4); break;
    }
    KC_HIP(ctx, hipGetLastError());
    u64 mx = 0;
    KC_HIP(ctx, hipMemcpyAsync(&mx, set->counters + 4, 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    set->nstride = kcs_stride_for(mx);
    set->dense_ok = true;
    return KC_OK;
}

// hash-sharded sets: every rank prepares its own blocks, then the ranks agree on the ids per block (the largest block anywhere), so
// that node id / nstride is the GLOBAL block and says which rank holds the node.  Collective.
int kcs_prepare_dense_sharded(kc_set *set, kc_comm *c) {
    KC_TRY(kcs_prepare_dense(set));
    u64 mx = set->nstride;
    KC_TRY(kcc_allreduce_max(c, &mx, 1));
    set->nstride = (u32)mx;
    return KC_OK;
}

int kcs_refresh_size(kc_set *set) {
    kc_ctx *ctx = set->ctx;
    u64 h[2] = {0, 0};
    KC_HIP(ctx, hipMemcpyAsync(h, set->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    set->size_host = h[0];
    if (h[1]) return kc_fail(ctx, KC_E_CAPACITY, "k-mer set capacity exhausted");
    return KC_OK;
}

template <int W> static SetView<W> view_of(const kc_set *s) { return kc_view<W>(s); }

// Move every k-mer (min_count == 0) or every k-mer counted >= min_count times into the freshly allocated table.
static int kcs_rebuild(kc_set *set, u64 ncap, u64 nnblocks, u64 *nkeys, u32 *nstate, u8 *ntags, u32 min_count) {
    kc_ctx *ctx = set->ctx;
    kcs_void_dense(set);
    kc_set to = *set;
    to.keys = nkeys; to.state = nstate; to.tags = ntags; to.cap = ncap; to.nblocks = nnblocks;
    u32 *ncounts = nullptr;
    if (set->counts && min_count == 0) {   // growing while counting: the counters move with their k-mers
        KC_TRY(kc_alloc(ctx, &ncounts, ncap));
        KC_HIP(ctx, hipMemsetAsync(ncounts, 0, ncap * 4, ctx->stream));
    }
    if (set->size_host) {
        const dim3 g = kc_grid2(set->cap, 256);
        switch (set->W) {
            case 1: KC_LAUNCH(kc_rehash_kernel<1>, dim3(g), dim3(256), 0, ctx->stream, view_of<1>(set), view_of<1>(&to), set->counters, (const u32 *)set->counts, ncounts, min_count); break;
            case 2: KC_LAUNCH(kc_rehash_kernel<2>, dim3(g), dim3(256), 0, ctx->stream, view_of<2>(set), view_of<2>(&to), set->counters, (const u32 *)set->counts, ncounts, min_count); break;
            default: KC_LAUNCH(kc_rehash_kernel<4>, dim3(g), dim3(256), 0, ctx->stream, view_of<4>(set), view_of<4>(&to), set->counters, (const u32 *)set->counts, ncounts, min_count); break;
        }
        KC_HIP(ctx, hipGetLastError());
    }
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, set->keys);
    kc_release(ctx, set->state);
    kc_release(ctx, set->tags);
    kc_release(ctx, set->counts);
    set->keys = nkeys; set->state = nstate; set->tags = ntags; set->counts = ncounts; set->cap = ncap; set->nblocks = nnblocks;
    return KC_OK;
}

// Make room for `additional` more k-mer occurrences: grow + rehash when the load could exceed 0.7.
int kcs_reserve(kc_set *set, u64 additional) {
    kc_ctx *ctx = set->ctx;
    KC_TRY(kcs_refresh_size(set));
    u64 worst = set->size_host + additional;
    if ((double)worst <= 0.7 * (double)set->cap) return KC_OK;
    u64 ncap = (u64)((double)worst / 0.6) + 1024;
    if (ncap < set->cap * 2) ncap = set->cap * 2;
    u64 nnblocks = (ncap + (((u64)1 << set->bshift) - 1)) >> set->bshift;
    ncap = nnblocks << set->bshift;
    u64 *nkeys = nullptr; u32 *nstate = nullptr; u8 *ntags = nullptr;
    KC_TRY(set_alloc_table(set, ncap, &nkeys, &nstate, &ntags));
    return kcs_rebuild(set, ncap, nnblocks, nkeys, nstate, ntags, 0);
}

static int insert_direct(kc_set *set, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int case_sensitive,
                         const u64 *only_over = nullptr, u64 over_limit = 0) {
    kc_ctx *ctx = set->ctx;
    int ppt = set->W == 1 ? 16 : (set->W == 2 ? 32 : 64);
    u64 nthreads = (len + ppt - 1) / ppt;
    const int win = kc_roll_win(set->k, set->mlen);
    const unsigned bs = win <= 32 ? 256 : (win <= 64 ? 128 : 64);    // LDS: 2 * win * bs * 4 bytes (<= 64 KB)
    const size_t lds = (size_t)win * bs * 4 * 2;   // raw + suffix-minimum arrays of kc_roll_kmers
    unsigned g = kc_grid(nthreads, bs);
    switch (set->W) {
        case 1: KC_LAUNCH(kc_insert_kernel<1>, dim3(g), dim3(bs), lds, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements, case_sensitive, view_of<1>(set), set->counters, ppt, set->counts, only_over, over_limit); break;
        case 2: KC_LAUNCH(kc_insert_kernel<2>, dim3(g), dim3(bs), lds, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements, case_sensitive, view_of<2>(set), set->counters, ppt, set->counts, only_over, over_limit); break;
        default: KC_LAUNCH(kc_insert_kernel<4>, dim3(g), dim3(bs), lds, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements, case_sensitive, view_of<4>(set), set->counters, ppt, set->counts, only_over, over_limit); break;
    }
    KC_HIP(ctx, hipGetLastError());
    return KC_OK;
}

// bucketed build: see kc_bucket_scatter_kernel.  Needs minimizer blocks whose table fits the LDS of a CU (16384 / 8192 /
// 4096 slots for 1 / 2 / 4 key words), no occurrence counting; small inputs are not worth the three launches.
static bool bucketed_applies(const kc_set *set, u64 len) {
    u64 min_len = (u64)1 << 20;
    if (const char *e = getenv("KC_BUCKET_MIN")) min_len = (u64)atoll(e);   // tests force either path
    return set->mlen && set->bshift <= bucketed_max_bshift(set->W) && set->bshift >= 9 && !set->counts && len >= min_len &&
           set->local_probes + 8 < ((u32)1 << set->bshift) / 2;
}

template <int W> static int insert_bucketed_w(kc_set *set, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int case_sensitive) {
    kc_ctx *ctx = set->ctx;
    KcScope scope(ctx);
    const u64 bs = (u64)1 << set->bshift;
    const u64 ovf_cap = (u64)1 << 20;
    u64 *bucket = nullptr, *bcursor = nullptr, *ovf = nullptr;
    KC_TRY(scope.alloc(&bucket, set->cap * W));
    KC_TRY(scope.alloc(&bcursor, set->nblocks));
    KC_TRY(scope.alloc(&ovf, ovf_cap * W));
    // keys beyond a full bucket (kc_scatter_aside; the wave-cooperative scatters): a list of their own, inserted with the spilled keys
    const u64 sovf_cap = (u64)1 << 16;
    u64 *sovf = nullptr;
    KC_TRY(scope.alloc(&sovf, sovf_cap * W));
    KC_HIP(ctx, hipMemsetAsync(bcursor, 0, set->nblocks * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 16, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 5, 0, 8, ctx->stream));
    const int win = kc_roll_win(set->k, set->mlen);
    // positions per thread: every thread first rolls over the k-1 bases before its range, so wide k wants long ranges
    // (every thread first rolls over the k - 1 bases before its range: longer ranges waste less of that, shorter ones give
    // small inputs more threads; measured at 2.5e9 31-mers: 64.0 / 60.5 / 59.2 ms for 16 / 32 / 64 positions)
    int ppt = W == 1 ? (len >= ((u64)1 << 28) ? 32 : 16) : (W == 2 ? 32 : 64);
    if (const char *e = getenv("KC_PPT")) { int v = atoi(e); if (v >= 8 && v <= 512) ppt = v; }   // tuning knob
    const u32 stage_cap = 8;                                           // staged keys per thread (a longer run is flushed in pieces)
    const unsigned tb = win <= 32 ? 256 : (win <= 64 ? 128 : 64);    // LDS: 2 * win * tb * 4 bytes (<= 64 KB) + staged run keys
    const size_t lds_scatter = (size_t)win * tb * 4 * 2 + (size_t)stage_cap * W * tb * 8;
    const u64 nthreads = (len + ppt - 1) / ppt;
    // (set on every call: the attribute belongs to the current device's copy of the kernel, and the call is cheap)
    if (lds_scatter > 64 * 1024)
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_bucket_scatter_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
    bool wave_form = false;
    if constexpr (W == 1) {
        // the wave-cooperative form (kc_bucket_scatter_wave_kernel): single-word keys on a minimizer table; KC_SCATTER_ROLL=1 keeps the rolling form
        wave_form = set->mlen && set->mlen <= 16 && set->k - set->mlen - 1 >= 1 && set->k - set->mlen - 1 <= 15 && !getenv("KC_SCATTER_ROLL");
        if (wave_form) {
            int chunks = 16;   // (even)
            if (const char *e = getenv("KC_SCATTER_CHUNKS")) { const int v = atoi(e); if (v >= 2 && v <= 4096) chunks = v & ~1; }   // tuning knob
            const u64 nwaves = (len + 64 * (u64)chunks - 1) / (64 * (u64)chunks);
            int exp_ = 0;
            if (const char *e = KC_EXP_KNOB("KC_SCATTER_EXP")) exp_ = atoi(e) & 3;   // (timing experiments: see the kernel)
            if (getenv("KC_SCATTER_V1") || exp_)
            KC_LAUNCH(kc_bucket_scatter_wave_kernel, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements,
                               case_sensitive, kc_view_raw<1>(set), bucket, bcursor, set->counters, chunks | (exp_ << 16), sovf, sovf_cap);
            else
            KC_LAUNCH(kc_bucket_scatter_wave2_kernel, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements,
                               case_sensitive, kc_view_raw<1>(set), bucket, bcursor, set->counters, chunks, sovf, sovf_cap);
        }
    }
    if constexpr (W > 1) {
        wave_form = set->mlen && set->mlen <= 16 && set->k - set->mlen - 1 >= 1 && set->k - set->mlen - 1 <= 127 && !getenv("KC_SCATTER_ROLL");
        if (wave_form) {
            const int chunks = 16;
            const u64 nwaves = (len + 64 * (u64)chunks - 1) / (64 * (u64)chunks);
            KC_LAUNCH(kc_bucket_scatter_wave_wide_kernel<W>, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k,
                               set->complements, case_sensitive, kc_view_raw<W>(set), bucket, bcursor, set->counters, chunks, sovf, sovf_cap);
        }
    }
    if (!wave_form)
        KC_LAUNCH(kc_bucket_scatter_kernel<W>, dim3(kc_grid(nthreads, tb)), dim3(tb), lds_scatter, ctx->stream, d_codes, d_brk, d_low, len,
                           set->k, set->complements, case_sensitive, kc_view_raw<W>(set), bucket, bcursor, set->counters, ppt, stage_cap, (const u64 *)nullptr, 0);
    const size_t lds_build = (size_t)bs * (8 * W + 1);
    KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_block_build_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
    const int fresh = set->size_host == 0 ? 1 : 0;
    // The table was sized by k-mer OCCURRENCES (the number of distinct k-mers is unknown before the build; khash grows with the
    // distinct keys, khash.h:314-321).  On a fresh set, count the distinct keys of a sample of the scatter buckets (a bucket holds
    // ALL occurrences of its k-mers, so the sample is exact for its part of the k-mer space); when every k-mer occurs r >= 2 times
    // on average (reads with coverage), build a table with r times fewer blocks, each from r neighbouring buckets -- every
    // table-sized pass of phase 1 then costs what the distinct k-mers cost, not what the occurrences cost.
    int merge_r = 0;
    u64 scatter_ovf = 0;
    if (fresh && !set->rcap && set->nblocks >= 256 && !getenv("KC_NO_RESIZE")) {
        KC_HIP(ctx, hipMemcpyAsync(&scatter_ovf, set->counters + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // (a bucket that overflowed keeps only part of its keys, and the clean-up sweep below addresses blocks of the scatter
        // geometry: such inputs keep the table as it is)
        if (!scatter_ovf) {
            const int stride = set->nblocks >= 64 * 128 ? 64 : (int)(set->nblocks / 128);
            const u64 nsample = (set->nblocks + stride - 1) / stride;
            KC_LAUNCH(kc_block_build_kernel<W>, dim3((unsigned)nsample), dim3(1024), lds_build, ctx->stream, kc_view_raw<W>(set), (const u64 *)bucket,
                               (const u64 *)bcursor, 1, set->counters, ovf, ovf_cap, (const u64 *)nullptr, 0, 0, stride);
            u64 hs[4] = {0, 0, 0, 0};
            KC_HIP(ctx, hipMemcpyAsync(hs, set->counters, sizeof hs, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 16, ctx->stream));
            const u64 distinct = hs[2], occ = hs[3];
            set->occ_ratio = distinct ? (float)((double)occ / (double)distinct) : 0.f;
            int r = 1;
            while (2 * r <= KC_MAX_SEG && distinct && occ >= (u64)(2 * r) * distinct && set->nblocks % (u64)(2 * r) == 0 && set->nblocks / (u64)(2 * r) >= 64) r *= 2;
            if (getenv("KC_SET_TRACE"))
                fprintf(stderr, "[set] sample of %llu buckets: %llu occurrences, %llu distinct -> %d scatter bucket(s) per table block\n", (unsigned long long)nsample,
                        (unsigned long long)occ, (unsigned long long)distinct, r);
            if (r > 1) {
                merge_r = r;
                kc_release(ctx, set->keys); kc_release(ctx, set->state); kc_release(ctx, set->tags);
                set->keys = nullptr; set->state = nullptr; set->tags = nullptr;
                set->nblocks /= (u64)r;
                set->cap = set->nblocks << set->bshift;
                KC_TRY(set_alloc_table(set, set->cap, &set->keys, &set->state, &set->tags, false));
            }
        }
    }
    if (!fresh) KC_TRY(kcs_clean_table(set));   // (cannot be pending on a non-empty set; kept for clarity)
    set->unclean = false;                        // the build kernel writes every slot, state and tag of every block
    // the blocks leave the build with their dense node ids prepared (void again if anything is inserted behind the kernel's back)
    kcs_void_dense(set);
    KC_TRY(kc_alloc(ctx, &set->ncount, set->nblocks + 2));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 4, 0, 8, ctx->stream));
    if (getenv("KC_NO_BUILD_IDS"))   // (test knob: the sweep of kcs_prepare_dense instead)
        KC_LAUNCH(kc_block_build_kernel<W>, dim3((unsigned)set->nblocks), dim3(1024), lds_build, ctx->stream, kc_view_raw<W>(set), (const u64 *)bucket,
                           (const u64 *)bcursor, fresh | (getenv("KC_BUILD_NOPREFETCH") ? 2 : 0) | (getenv("KC_BUILD_ST8") ? 4 : 0) | (KC_EXP_KNOB("KC_BUILD_EXP") ? (atoi(KC_EXP_KNOB("KC_BUILD_EXP")) & 7) << 4 : 0), set->counters, ovf, ovf_cap, (const u64 *)nullptr, 0, merge_r, 0, (u32 *)nullptr);
    else {
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_block_build_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
        KC_LAUNCH((kc_block_build_kernel<W, true>), dim3((unsigned)set->nblocks), dim3(1024), lds_build, ctx->stream, kc_view_raw<W>(set), (const u64 *)bucket,
                           (const u64 *)bcursor, fresh | (getenv("KC_BUILD_NOPREFETCH") ? 2 : 0) | (getenv("KC_BUILD_ST8") ? 4 : 0) | (KC_EXP_KNOB("KC_BUILD_EXP") ? (atoi(KC_EXP_KNOB("KC_BUILD_EXP")) & 7) << 4 : 0), set->counters, ovf, ovf_cap, (const u64 *)nullptr, 0, merge_r, 0, set->ncount);
        KC_LAUNCH(kc_max_u32_kernel, dim3(64), dim3(256), 0, ctx->stream, (const u32 *)set->ncount, set->nblocks, set->counters + 4);
    }
    KC_HIP(ctx, hipGetLastError());
    u64 h[6] = {0, 0, 0, 0, 0, 0};
    KC_HIP(ctx, hipMemcpyAsync(h, set->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h[3] |= scatter_ovf;
    const u64 aside = h[3] ? 0 : (h[5] < sovf_cap ? h[5] : sovf_cap);   // keys beyond a full bucket (with the flag up the second sweep takes those blocks whole)
    if (!h[2] && !h[3] && !aside && !getenv("KC_NO_BUILD_IDS")) { set->nstride = kcs_stride_for(h[4]); set->dense_ok = true; }   // (KC_NO_BUILD_IDS: test knob, the sweep of kcs_prepare_dense instead)
    set->homes_primary = fresh && !h[2] && !h[3] && !aside;   // no key left its block (spills and keys beyond a full bucket take the global insert)
    if (getenv("KC_SET_TRACE"))
        fprintf(stderr, "[set] bucketed build W=%d blocks=%llu (2^%d slots): size %llu, spills %llu, bucket overflow %llu (%llu keys aside), largest block %llu k-mers\n", W,
                (unsigned long long)set->nblocks, set->bshift, (unsigned long long)h[0], (unsigned long long)h[2], (unsigned long long)h[3], (unsigned long long)h[5],
                (unsigned long long)h[4]);
    if (h[2] > ovf_cap)   // the spill list overflowed: the global path inserts whatever is still missing
        return insert_direct(set, d_codes, d_brk, d_low, len, case_sensitive);
    if (h[3]) {           // some buckets overflowed (blocks hit by high-multiplicity repeats): a second sweep inserts only those blocks' k-mers
        KC_TRY(insert_direct(set, d_codes, d_brk, d_low, len, case_sensitive, bcursor, bs));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (bcursor belongs to this scope)
    }
    if (h[2] || aside) {
        // the spilled keys go to their second homes by the global insert; the build kernel has numbered the blocks' nodes already, so the
        // few blocks that take a key now are numbered again (k = 127 on 10^8 reads: 254 spills -- the sweep over the whole table that a
        // voided numbering costs was 3.6 ms of a 27 ms pass)
        // (a renumbering workgroup rewrites a whole block: with spills in more than 1/16 of the blocks the one sweep of kcs_prepare_dense,
        // which phase 1 runs when dense_ok is unset, is cheaper than block-by-block repairs)
        const u64 nblk_here = (set->rcap ? set->rcap : set->cap) >> set->bshift;
        const u64 nlist = h[2] + aside;
        const bool renumber = !h[3] && !getenv("KC_NO_BUILD_IDS") && set->ncount && fresh && nlist * 16 <= nblk_here;
        u64 *touched = nullptr;
        u32 *claim = nullptr;
        auto rel = kc_defer([&] { if (touched || claim) { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, touched); kc_release(ctx, claim); } });
        if (renumber) {
            KC_TRY(kc_alloc(ctx, &touched, nlist));
            KC_TRY(kc_alloc(ctx, &claim, nblk_here));
            KC_HIP(ctx, hipMemsetAsync(claim, 0, nblk_here * sizeof(u32), ctx->stream));
        }
        if (h[2])
            KC_LAUNCH(kc_insert_list_kernel<W>, dim3(kc_grid(h[2], 256)), dim3(256), 0, ctx->stream, view_of<W>(set), (const u64 *)ovf, h[2],
                               set->counters, touched);
        if (aside)
            KC_LAUNCH(kc_insert_list_kernel<W>, dim3(kc_grid(aside, 256)), dim3(256), 0, ctx->stream, view_of<W>(set), (const u64 *)sovf, aside,
                               set->counters, touched ? touched + h[2] : (u64 *)nullptr);
        KC_HIP(ctx, hipGetLastError());
        if (renumber) {
            KC_LAUNCH(kc_annotate_kernel<W>, dim3((unsigned)nlist), dim3(256), 0, ctx->stream, kc_view_raw<W>(set), set->ncount, set->counters + 4, (const u64 *)touched, claim);
            KC_HIP(ctx, hipGetLastError());
            u64 h2[5] = {0, 0, 0, 0, 0};
            KC_HIP(ctx, hipMemcpyAsync(h2, set->counters, sizeof h2, hipMemcpyDeviceToHost, ctx->stream));
            KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!h2[1]) { set->nstride = kcs_stride_for(h2[4]); set->dense_ok = true; }   // (counters[1]: the table is full -- the caller's problem, not a numbering)
        }
    }
    return KC_OK;
}
static int insert_bucketed(kc_set *set, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int case_sensitive) {
    switch (set->W) {
        case 1: return insert_bucketed_w<1>(set, d_codes, d_brk, d_low, len, case_sensitive);
        case 2: return insert_bucketed_w<2>(set, d_codes, d_brk, d_low, len, case_sensitive);
        default: return insert_bucketed_w<4>(set, d_codes, d_brk, d_low, len, case_sensitive);
    }
}

// ------------------------------------------------------------------------------------------------- hash-sharded set
// The table geometry (block size, minimizer length, load) is that of kc_set_create for the TOTAL number of occurrences;
// the number of blocks is rounded up to a multiple of the world size and rank r owns blocks [r * bpr, (r + 1) * bpr).
struct ShardGeom { int W, mlen, bshift; u32 local_probes; u64 nblocks, bpr; };
static ShardGeom shard_geometry(int k, u64 occ, int world, double load) {
    ShardGeom g;
    g.W = kc_words_for_k(k); g.mlen = kc_default_mlen(k); g.bshift = g.W == 4 ? 12 : 13; g.local_probes = 128;
    if (const char *e = getenv("KC_BSHIFT")) { int v = atoi(e); if (v >= 9 && v <= bucketed_max_bshift(g.W)) g.bshift = v; }
    if (const char *e = getenv("KC_LOCAL_PROBES")) { int v = atoi(e); if (v >= 1) g.local_probes = (u32)v; }
    if (g.local_probes + 8 >= ((u32)1 << g.bshift) / 2) g.local_probes = ((u32)1 << g.bshift) / 4;
    const u64 bs = (u64)1 << g.bshift;
    const u64 cap = (u64)((double)occ / load) + 1024;
    u64 nblocks = (cap + bs - 1) >> g.bshift;
    nblocks = (nblocks + world - 1) / world * world;
    g.nblocks = nblocks; g.bpr = nblocks / world;
    return g;
}
// Host-side description of the partition (needs no GPU): out[0] = table blocks in total, [1] = log2 slots per block,
// [2] = blocks per rank, [3] = first global slot of `rank`, [4] = slots per rank, [5] = minimizer length.
extern "C" int kc_shard_geometry(int k, uint64_t total_occurrences, int world, int rank, uint64_t *out6) {
    if (!out6 || k < 18 || k > 127 || world < 1 || rank < 0 || rank >= world) return KC_E_ARG;
    const ShardGeom g = shard_geometry(k, total_occurrences, world, 0.5);
    out6[0] = g.nblocks; out6[1] = (u64)g.bshift; out6[2] = g.bpr; out6[3] = ((u64)rank * g.bpr) << g.bshift; out6[4] = g.bpr << g.bshift; out6[5] = (u64)g.mlen;
    return KC_OK;
}
// owner rank of n canonical k-mers (W words each): the rank that holds the table block of the k-mer's canonical minimizer
template <int W> static void owner_of_w(int k, const ShardGeom &g, const u64 *keys, u64 n, u8 *owner) {
    SetView<W> v{nullptr, nullptr, nullptr, g.nblocks << g.bshift, g.nblocks, g.bshift, g.mlen, k, g.local_probes, 0, g.nblocks << g.bshift, 1};
    for (u64 i = 0; i < n; ++i) {
        Key<W> key;
        for (int w = 0; w < W; ++w) key.w[w] = keys[i * W + w];
        const u64 home = set_block_home(v, key_min_mmh(key, k, g.mlen, 1, k - g.mlen - 1), 0);
        owner[i] = (u8)((home >> g.bshift) / g.bpr);
    }
}
extern "C" int kc_shard_owner_of(int k, uint64_t total_occurrences, int world, const uint64_t *keys, uint64_t n, uint8_t *owner) {
    if ((!keys || !owner) && n) return KC_E_ARG;
    if (k < 18 || k > 127 || world < 1) return KC_E_ARG;
    const ShardGeom g = shard_geometry(k, total_occurrences, world, 0.5);
    switch (g.W) {
        case 1: owner_of_w<1>(k, g, keys, n, owner); break;
        case 2: owner_of_w<2>(k, g, keys, n, owner); break;
        default: owner_of_w<4>(k, g, keys, n, owner); break;
    }
    return KC_OK;
}

int kcs_create_sharded(kc_ctx *ctx, kc_comm *c, int k, int complements, u64 occ_local, kc_set **out) {
    *out = nullptr;
    if (k < 18) return kc_fail(ctx, KC_E_ARG, "the hash-sharded path needs k >= 18 (minimizer blocks)");
    u64 occ = occ_local;
    KC_TRY(kcc_allreduce_sum(c, &occ, 1));
    kc_set *s = new kc_set();
    s->ctx = ctx; s->k = k; s->W = kc_words_for_k(k); s->complements = complements ? 1 : 0;
    double load = 0.5;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
        double per_slot = 8.0 * s->W + 1 + (s->W > 1 ? 4 : 0) + 16 + 8;
        double need = (double)occ / c->world / load * per_slot + (double)occ / c->world * (16.0 * s->W + 60);
        if (total_b && need > 0.6 * (double)total_b) load = 0.66;
    }
    if (const char *e = getenv("KC_LOAD")) { double v = atof(e); if (v >= 0.2 && v <= 0.9) load = v; }
    const ShardGeom g = shard_geometry(k, occ, c->world, load);
    s->mlen = g.mlen; s->bshift = g.bshift; s->local_probes = g.local_probes;
    const u64 nblocks = g.nblocks, bpr = g.bpr;
    s->nblocks = nblocks; s->cap = nblocks << s->bshift;
    s->rcap = bpr << s->bshift; s->rlo = (u64)c->rank * s->rcap;
    int rc = kc_alloc(ctx, &s->keys, s->rcap * s->W);
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->tags, s->rcap + 8);
    if (rc == KC_OK && s->W != 1) rc = kc_alloc(ctx, &s->state, s->rcap);
    s->unclean = true;
    if (rc == KC_OK) rc = kc_alloc(ctx, &s->counters, 8);   // [0] size, [1] capacity flag, [2] [3] scratch of the builds, [4] largest block (dense ids)
    if (rc == KC_OK && hipMemsetAsync(s->counters, 0, 8 * sizeof(u64), ctx->stream) != hipSuccess) rc = KC_E_NO_DEVICE;
    if (rc != KC_OK) { kc_set_destroy(s); return rc; }
    *out = s;
    return KC_OK;
}

// AddKMers over all ranks: count the occurrences per table block, lay the canonical k-mers out block by block (= owner by
// owner), all-to-all the per-block counts and the keys, build every owned block in LDS from its `world` segments.
// dense layout of the scatter buckets: bucket b's first cnt[b] keys -> out[off[b] ...] (one workgroup per bucket)
__global__ void __launch_bounds__(256) kc_bucket_compact_kernel(const u64 *__restrict__ fixed, const u64 *__restrict__ cnt, const u64 *__restrict__ off, int bshift, int W,
                                                                u64 *out) {
    const u64 b = blockIdx.x;
    const u64 n = cnt[b] * (u64)W;
    const u64 *src = fixed + ((b << bshift) * (u64)W);
    u64 *dst = out + off[b] * (u64)W;
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

template <int W> static int build_sharded_w(kc_set *set, kc_comm *c, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, u64 *moved, int min_count) {
    kc_ctx *ctx = set->ctx;
    KcScope scope(ctx);
    const int G = c->world;
    const u64 nblocks = set->nblocks, bpr = nblocks / G, bs = (u64)1 << set->bshift;
    u64 *bcount = nullptr, *boff = nullptr, *bcur = nullptr, *bucket = nullptr, *rcounts = nullptr, *seg_scan = nullptr, *rkeys = nullptr, *ovf = nullptr;
    KC_TRY(scope.alloc(&bcount, nblocks + 1)); KC_TRY(scope.alloc(&boff, nblocks + 1)); KC_TRY(scope.alloc(&bcur, nblocks + 1));
    KC_HIP(ctx, hipMemsetAsync(bcount, 0, (nblocks + 1) * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(bcur, 0, (nblocks + 1) * 8, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 16, ctx->stream));
    const int win = kc_roll_win(set->k, set->mlen);
    const int ppt = W == 1 ? 16 : (W == 2 ? 32 : 64);
    const u32 stage_cap = 8;
    const unsigned tb = win <= 32 ? 256 : (win <= 64 ? 128 : 64);
    const size_t lds_scatter = (size_t)win * tb * 4 * 2 + (size_t)stage_cap * W * tb * 8;
    const u64 nthreads = (len + ppt - 1) / ppt;
    if (lds_scatter > 64 * 1024)
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_bucket_scatter_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
    const bool have = len >= (u64)set->k;
    // ONE roll over the input: the k-mers go into fixed-capacity buckets (one of 2^bshift keys per table block, as on one GPU) and a
    // copy lays them out densely by block = by owner afterwards.  (Counting first and scattering into the exact layout rolled
    // every k-mer twice: +12 ms per 10^9.)  A bucket that overflows (a block that receives more than 2^bshift occurrences from
    // this rank alone) sends the rank back to the two-pass form.
    u64 *fixed = nullptr;
    bool counted = false;   // bcount holds the per-block counts already
    bool one_pass = have && !getenv("KC_SHARD_TWO_PASS");
    if (one_pass) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
        if ((double)nblocks * (double)bs * 8.0 * W > 0.5 * (double)free_b) one_pass = false;
    }
    if (one_pass) {
        KC_TRY(scope.alloc(&fixed, nblocks * bs * W));
        bool wave_form = false;
        if constexpr (W == 1) {   // (as in insert_bucketed_w)
            wave_form = set->mlen && set->mlen <= 16 && set->k - set->mlen - 1 >= 1 && set->k - set->mlen - 1 <= 15 && !getenv("KC_SCATTER_ROLL");
            if (wave_form) {
                const int chunks = 16;
                const u64 nwaves = (len + 64 * (u64)chunks - 1) / (64 * (u64)chunks);
                if (getenv("KC_SCATTER_V1"))
                KC_LAUNCH(kc_bucket_scatter_wave_kernel, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements,
                                   0, kc_view_raw<1>(set), fixed, bcount, set->counters, chunks);
                else
                KC_LAUNCH(kc_bucket_scatter_wave2_kernel, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements,
                                   0, kc_view_raw<1>(set), fixed, bcount, set->counters, chunks);
            }
        }
        if constexpr (W > 1) {
            wave_form = set->mlen && set->mlen <= 16 && set->k - set->mlen - 1 >= 1 && set->k - set->mlen - 1 <= 127 && !getenv("KC_SCATTER_ROLL");
            if (wave_form) {
                const int chunks = 16;
                const u64 nwaves = (len + 64 * (u64)chunks - 1) / (64 * (u64)chunks);
                KC_LAUNCH(kc_bucket_scatter_wave_wide_kernel<W>, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k,
                                   set->complements, 0, kc_view_raw<W>(set), fixed, bcount, set->counters, chunks);
            }
        }
        if (!wave_form)
            KC_LAUNCH(kc_bucket_scatter_kernel<W>, dim3(kc_grid(nthreads, tb)), dim3(tb), lds_scatter, ctx->stream, d_codes, d_brk, d_low, len, set->k,
                               set->complements, 0, kc_view_raw<W>(set), fixed, bcount, set->counters, ppt, stage_cap, (const u64 *)nullptr, 0);
        u64 hovf = 0;
        KC_HIP(ctx, hipMemcpyAsync(&hovf, set->counters + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hovf) {
            // (the cursors of the failed pass ARE the per-block counts -- every run was reserved, whether it fitted or not: no counting
            // pass; round 5, `-z` on reads with coverage, where blocks of twice the mean are common: 26 ms per call)
            one_pass = false;
            counted = wave_form;
            scope.release(fixed); fixed = nullptr;
            if (!counted) KC_HIP(ctx, hipMemsetAsync(bcount, 0, (nblocks + 1) * 8, ctx->stream));
            KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 16, ctx->stream));
        }
    }
    if (have && !one_pass && !counted)
        KC_LAUNCH(kc_bucket_scatter_kernel<W>, dim3(kc_grid(nthreads, tb)), dim3(tb), lds_scatter, ctx->stream, d_codes, d_brk, d_low, len, set->k,
                           set->complements, 0, kc_view_raw<W>(set), (u64 *)nullptr, bcount, set->counters, ppt, stage_cap, (const u64 *)nullptr, 1);
    KC_TRY(kcu_exclusive_scan(ctx, bcount, boff, nblocks));
    std::vector<u64> h_off(nblocks + 1);
    KC_HIP(ctx, hipMemcpyAsync(h_off.data(), boff, (nblocks + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const u64 occ_local = h_off[nblocks];
    KC_TRY(scope.alloc(&bucket, occ_local * W + 1));
    if (one_pass) {
        KC_LAUNCH(kc_bucket_compact_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->stream, (const u64 *)fixed, (const u64 *)bcount, (const u64 *)boff, set->bshift, W, bucket);
        KC_HIP(ctx, hipGetLastError());
        KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        scope.release(fixed);
    } else if (have) {
        bool by_wave = false;
        if constexpr (W == 1) {   // the exact layout by the wave-cooperative scatter where it applies (17 against 26 ms per 2.5 * 10^9 positions)
            by_wave = set->mlen == 16 && set->k >= 18 && set->k <= 32 && !getenv("KC_SCATTER_ROLL") && !getenv("KC_SCATTER_V1");
            // (that kernel hands a run's base on in 32 bits: a block with 2^32 occurrences or more -- some 4.3 * 10^9 copies of one k-mer -- is
            // left to the rolling form)
            for (u64 b2 = 0; by_wave && b2 < nblocks; ++b2) by_wave = h_off[b2 + 1] - h_off[b2] < ((u64)1 << 32);
            if (by_wave) {
                const int chunks = 16;
                const u64 nwaves = (len + 64 * (u64)chunks - 1) / (64 * (u64)chunks);
                KC_LAUNCH(kc_bucket_scatter_wave2_kernel, dim3(kc_grid(nwaves * 64, 256)), dim3(256), 0, ctx->stream, d_codes, d_brk, d_low, len, set->k, set->complements,
                                   0, kc_view_raw<1>(set), bucket, bcur, set->counters, chunks, (u64 *)nullptr, (u64)0, (const u64 *)boff);
            }
        }
        if (!by_wave)
            KC_LAUNCH(kc_bucket_scatter_kernel<W>, dim3(kc_grid(nthreads, tb)), dim3(tb), lds_scatter, ctx->stream, d_codes, d_brk, d_low, len, set->k,
                               set->complements, 0, kc_view_raw<W>(set), bucket, bcur, set->counters, ppt, stage_cap, (const u64 *)boff, 0);
    }
    KC_HIP(ctx, hipGetLastError());
    // exchange 1: the per-block counts of every owner's blocks
    std::vector<u64> so(G), sc(G), ro(G), rcv(G);
    KC_TRY(scope.alloc(&rcounts, (u64)G * bpr + 1));
    for (int p = 0; p < G; ++p) { so[p] = (u64)p * bpr * 8; sc[p] = bpr * 8; ro[p] = (u64)p * bpr * 8; rcv[p] = bpr * 8; }
    KC_TRY(kcc_alltoallv(c, bcount, so.data(), sc.data(), rcounts, ro.data(), rcv.data()));
    // exchange 2: the keys themselves
    std::vector<u64> totals(G), all((size_t)G * G);
    for (int p = 0; p < G; ++p) totals[p] = h_off[(u64)(p + 1) * bpr] - h_off[(u64)p * bpr];
    KC_TRY(kcc_allgather_words(c, totals.data(), all.data(), G));
    u64 nrecv = 0;
    for (int p = 0; p < G; ++p) {
        so[p] = h_off[(u64)p * bpr] * W * 8; sc[p] = totals[p] * W * 8;
        ro[p] = nrecv * W * 8; rcv[p] = all[(size_t)p * G + c->rank] * W * 8;
        nrecv += all[(size_t)p * G + c->rank];
    }
    KC_TRY(scope.alloc(&rkeys, nrecv * W + 1));
    KC_TRY(kcc_alltoallv(c, bucket, so.data(), sc.data(), rkeys, ro.data(), rcv.data()));
    if (moved) *moved = (occ_local - totals[c->rank]) * W * 8;
    KC_TRY(scope.alloc(&seg_scan, (u64)G * bpr + 2));
    KC_TRY(kcu_exclusive_scan(ctx, rcounts, seg_scan, (u64)G * bpr));
    scope.release(bucket);
    // build the owned blocks
    const u64 ovf_cap = (u64)1 << 20;
    KC_TRY(scope.alloc(&ovf, ovf_cap * W));
    const size_t lds_build = (size_t)bs * (8 * W + 1) + (min_count > 1 ? 2 * (size_t)bs : 0);
    set->unclean = false;
    bool ids_built = false;
    if (min_count > 1) {
        // -z: count the occurrences of every key where they all meet, keep those seen min_count times or more
        unsigned short *where = nullptr;
        KC_TRY(scope.alloc(&where, nrecv + 4));
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_block_build_filter_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
        KC_HIP(ctx, hipMemsetAsync(set->counters + 3, 0, 8, ctx->stream));
        KC_LAUNCH(kc_block_build_filter_kernel<W>, dim3((unsigned)bpr), dim3(1024), lds_build, ctx->stream, kc_view_raw<W>(set), (const u64 *)rkeys, set->counters,
                           (const u64 *)seg_scan, G, (u32)min_count, where);
    } else {
        // (the blocks leave the build with their dense node ids prepared, as on one GPU: kcs_prepare_dense_sharded then only has to
        // agree on the largest block among the ranks)
        KC_TRY(kc_alloc(ctx, &set->ncount, bpr + 2));
        KC_HIP(ctx, hipMemsetAsync(set->counters + 4, 0, 8, ctx->stream));
        KC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kc_block_build_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, KC_MAX_DYN_LDS));
        KC_LAUNCH((kc_block_build_kernel<W, true>), dim3((unsigned)bpr), dim3(1024), lds_build, ctx->stream, kc_view_raw<W>(set), (const u64 *)rkeys,
                           (const u64 *)nullptr, 1, set->counters, ovf, ovf_cap, (const u64 *)seg_scan, G, 0, 0, set->ncount);
        KC_LAUNCH(kc_max_u32_kernel, dim3(64), dim3(256), 0, ctx->stream, (const u32 *)set->ncount, bpr, set->counters + 4);
        ids_built = true;
    }
    KC_HIP(ctx, hipGetLastError());
    u64 h[5] = {0, 0, 0, 0, 0};
    KC_HIP(ctx, hipMemcpyAsync(h, set->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ids_built && !h[2]) { set->nstride = kcs_stride_for(h[4]); set->dense_ok = true; }   // (spills are inserted behind the kernel's back: the sweep of kcs_prepare_dense then)
    set->homes_primary = ids_built && !h[2];   // (a fresh build whose keys all stayed in their blocks: the probe kernel's in-block guess applies on this rank)
    if (min_count > 1) {   // (collective: a rank whose blocks overflowed takes everybody out of the sharded path)
        u64 lost = h[3];
        KC_TRY(kcc_allreduce_max(c, &lost, 1));
        if (lost) return kc_fail_agreed(ctx, KC_E_UNSUPPORTED_SHARDED, "-z with table blocks that overflow is not part of the hash-sharded path");
    }
    if (getenv("KC_SET_TRACE"))
        fprintf(stderr, "[set] rank %d/%d sharded build W=%d: %llu of %llu blocks (2^%d slots), %llu occurrences in, %llu received, size %llu, spills %llu\n", c->rank, G, W,
                (unsigned long long)bpr, (unsigned long long)nblocks, set->bshift, (unsigned long long)occ_local, (unsigned long long)nrecv, (unsigned long long)h[0],
                (unsigned long long)h[2]);
    {   // (agreed among the ranks: one that returned alone here would leave its peers waiting in the next exchange)
        u64 over = h[2] > ovf_cap ? 1 : 0;
        KC_TRY(kcc_allreduce_max(c, &over, 1));
        if (over) return kc_fail_agreed(ctx, KC_E_CAPACITY, "hash-sharded build: too many k-mers spilled out of their blocks (on some rank)");
    }
    if (h[2]) {
        KC_LAUNCH(kc_insert_list_kernel<W>, dim3(kc_grid(h[2], 256)), dim3(256), 0, ctx->stream, kc_view_raw<W>(set), (const u64 *)ovf, h[2], set->counters);
        KC_HIP(ctx, hipGetLastError());
    }
    return kcs_refresh_size(set);
}
int kcs_build_sharded(kc_set *set, kc_comm *c, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, u64 *moved, int min_count) {
    kcs_void_dense(set);
    if (min_count > 1 && (((size_t)1 << set->bshift) * (8 * (size_t)set->W + 3) > KC_MAX_DYN_LDS || set->bshift > 15))
        return kc_fail_agreed(set->ctx, KC_E_UNSUPPORTED_SHARDED, "-z with this table geometry is not part of the hash-sharded path");
    switch (set->W) {
        case 1: return build_sharded_w<1>(set, c, d_codes, d_brk, d_low, len, moved, min_count);
        case 2: return build_sharded_w<2>(set, c, d_codes, d_brk, d_low, len, moved, min_count);
        default: return build_sharded_w<4>(set, c, d_codes, d_brk, d_low, len, moved, min_count);
    }
}

// -z on one GPU: the counted build of the hash-sharded flow (kc_block_build_filter_kernel: occurrences counted in LDS, where the
// k-mers of a block meet) with a single-rank communicator, turned into an ordinary set -- the whole table is this rank's -- and moved into
// a table of the survivors' size when the filter left it nearly empty (a table sized for 30x the k-mers it holds makes every later
// sweep 30x too long).  KC_E_ARG / KC_E_CAPACITY with ctx->err_agreed set: not applicable here (blocks that overflow, geometry): the
// caller counts with the direct insert instead.
int kcs_build_counted(kc_ctx *ctx, int k, int complements, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int min_count, kc_set **out) {
    *out = nullptr;
    ctx->err_agreed = false;
    kc_comm *self = nullptr;
    KC_TRY(kc_comm_create_self(ctx, &self));
    auto guard = kc_defer([&] { kc_comm_destroy(self); });
    kc_set *set = nullptr;
    const u64 occ = len >= (u64)k ? len - k + 1 : 0;
    KC_TRY(kcs_create_sharded(ctx, self, k, complements, occ, &set));
    u64 moved = 0;
    int rc = kcs_build_sharded(set, self, d_codes, d_brk, d_low, len, &moved, min_count);
    if (rc != KC_OK) { kc_set_destroy(set); return rc; }
    set->rlo = 0; set->rcap = 0;   // (one rank: its slot range is the table)
    rc = kcs_refresh_size(set);
    if (rc == KC_OK && set->size_host && set->size_host * 4 < set->cap) {
        u64 ncap = set->size_host * 2 + 1024;
        u64 nnblocks = (ncap + (((u64)1 << set->bshift) - 1)) >> set->bshift;
        if (nnblocks >= 4096) nnblocks = (nnblocks + 63) / 64 * 64;
        ncap = nnblocks << set->bshift;
        u64 *nkeys = nullptr; u32 *nstate = nullptr; u8 *ntags = nullptr;
        rc = set_alloc_table(set, ncap, &nkeys, &nstate, &ntags);
        if (rc == KC_OK) rc = kcs_rebuild(set, ncap, nnblocks, nkeys, nstate, ntags, 0);
        if (rc == KC_OK) rc = kcs_refresh_size(set);
    }
    if (rc != KC_OK) { kc_set_destroy(set); return rc; }
    *out = set;
    return KC_OK;
}

int kcs_insert_packed(kc_set *set, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int case_sensitive) {
    kc_ctx *ctx = set->ctx;
    if (len == 0) return KC_OK;
    kcs_void_dense(set);
    (void)hipEventRecord(ctx->ev[6], ctx->stream);
    int rc = bucketed_applies(set, len) ? insert_bucketed(set, d_codes, d_brk, d_low, len, case_sensitive)
                                        : insert_direct(set, d_codes, d_brk, d_low, len, case_sensitive);
    (void)hipEventRecord(ctx->ev[7], ctx->stream);
    return rc;
}

extern "C" int kc_set_add_sequence_device(kc_set *set, const void *d_seq, uint64_t len, int case_sensitive) {
    if (!set || (!d_seq && len)) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (len < (u64)set->k) return KC_OK;
    KC_TRY(kcs_reserve(set, len - set->k + 1));
    u64 *codes = nullptr, *brk = nullptr, *low = nullptr;
    int rc = kcs_pack(ctx, d_seq, len, &codes, &brk, &low);
    if (rc == KC_OK) rc = kcs_insert_packed(set, codes, brk, low, len, case_sensitive);
    if (rc == KC_OK) rc = kcs_refresh_size(set);
    kc_release(ctx, codes); kc_release(ctx, brk); kc_release(ctx, low);
    return rc;
}

extern "C" int kc_set_add_sequence(kc_set *set, const char *seq, uint64_t len, int case_sensitive) {
    if (!set || (!seq && len)) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (len < (u64)set->k) return KC_OK;
    u8 *d = nullptr;
    KC_TRY(kc_alloc(ctx, &d, len + 64));
    int rc = KC_OK;
    if (hipMemcpyAsync(d, seq, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = kc_fail(ctx, KC_E_NO_DEVICE, "H2D copy failed");
    if (rc == KC_OK) rc = kc_set_add_sequence_device(set, d, len, case_sensitive);
    (void)hipStreamSynchronize(ctx->stream);
    kc_release(ctx, d);
    return rc;
}

extern "C" int kc_set_enable_counting(kc_set *set) {
    if (!set) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    KC_TRY(kcs_refresh_size(set));
    if (set->size_host) return kc_fail(ctx, KC_E_ARG, "counting must be enabled on an empty set");
    if (!set->counts) KC_TRY(kc_alloc(ctx, &set->counts, set->cap));
    KC_HIP(ctx, hipMemsetAsync(set->counts, 0, set->cap * 4, ctx->stream));
    return KC_OK;
}

// kMersToVecFiltered + kMersFromVec (khash_utils.h:143-159): keep count >= min_frequency (counts saturate at 255 in the
// reference; min_frequency <= 255, so the unsaturated 32-bit counter gives the same decision)
extern "C" int kc_set_filter_min_frequency(kc_set *set, int min_frequency) {
    if (!set) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (min_frequency < 1 || min_frequency > 255) return kc_fail(ctx, KC_E_ARG, "Minimum frequency '-z' must be between 1 and 255.");
    if (!set->counts) return kc_fail(ctx, KC_E_ARG, "kc_set_enable_counting was not called");
    KC_TRY(kcs_refresh_size(set));
    u64 ncap = (u64)((double)set->size_host / 0.6) + 1024;
    u64 nnblocks = (ncap + (((u64)1 << set->bshift) - 1)) >> set->bshift;
    ncap = nnblocks << set->bshift;
    u64 *nkeys = nullptr; u32 *nstate = nullptr; u8 *ntags = nullptr;
    KC_TRY(set_alloc_table(set, ncap, &nkeys, &nstate, &ntags));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 8, ctx->stream));
    KC_TRY(kcs_rebuild(set, ncap, nnblocks, nkeys, nstate, ntags, (u32)min_frequency));
    u64 h[3];
    KC_HIP(ctx, hipMemcpyAsync(h, set->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h[0] = h[2];   // size = survivors
    KC_HIP(ctx, hipMemcpyAsync(set->counters, h, 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    set->size_host = h[2];
    return KC_OK;
}

// kh_del for a list of keys: flag the slots that hold them, then move every unflagged k-mer into a fresh table of the
// same size (the same rebuild as the -z filter: keep[slot] plays the occurrence counter, threshold 1).
template <int W>
__global__ void __launch_bounds__(256) kc_flag_erase_kernel(SetView<W> set, const u64 *__restrict__ keys, u64 n, u32 *keep, u64 *counter) {
    u64 t = kc_gtid();
    if (t >= n) return;
    Key<W> key;
    for (int i = 0; i < W; ++i) key.w[i] = keys[t * W + i];
    const u64 slot = set_find(set, key);
    if (slot == KC_NONE64) return;
    if (atomicExch(&keep[slot], 0u) == 1u) atomicAdd((ull *)counter, 1ULL);
}
__global__ void __launch_bounds__(256) kc_fill_u32_kernel(u32 *a, u64 n, u32 v) {
    u64 t = kc_gtid();
    if (t < n) a[t] = v;
}
extern "C" int kc_set_erase(kc_set *set, const uint64_t *key_words, uint64_t n, uint64_t *n_erased) {
    if (!set || (!key_words && n)) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    if (n_erased) *n_erased = 0;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (set->rcap) return kc_fail(ctx, KC_E_ARG, "kc_set_erase on a hash-sharded set");
    if (set->counts) return kc_fail(ctx, KC_E_ARG, "kc_set_erase while occurrences are being counted");
    KC_TRY(kcs_refresh_size(set));
    if (!n || !set->size_host) return KC_OK;
    u64 *d_keys = nullptr;
    u32 *keep = nullptr;
    KC_TRY(kc_alloc(ctx, &d_keys, n * set->W));
    int rc = kc_alloc(ctx, &keep, set->cap);
    if (rc != KC_OK) { kc_release(ctx, d_keys); return rc; }
    KC_HIP(ctx, hipMemcpyAsync(d_keys, key_words, n * set->W * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipMemsetAsync(set->counters + 2, 0, 16, ctx->stream));
    KC_LAUNCH(kc_fill_u32_kernel, kc_grid2(set->cap, 256), dim3(256), 0, ctx->stream, keep, set->cap, 1u);
    const dim3 g = kc_grid2(n, 256);
    switch (set->W) {
        case 1: KC_LAUNCH(kc_flag_erase_kernel<1>, g, dim3(256), 0, ctx->stream, view_of<1>(set), (const u64 *)d_keys, n, keep, set->counters + 3); break;
        case 2: KC_LAUNCH(kc_flag_erase_kernel<2>, g, dim3(256), 0, ctx->stream, view_of<2>(set), (const u64 *)d_keys, n, keep, set->counters + 3); break;
        default: KC_LAUNCH(kc_flag_erase_kernel<4>, g, dim3(256), 0, ctx->stream, view_of<4>(set), (const u64 *)d_keys, n, keep, set->counters + 3); break;
    }
    KC_HIP(ctx, hipGetLastError());
    u64 h[4];
    KC_HIP(ctx, hipMemcpyAsync(h, set->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, d_keys);
    const u64 erased = h[3];
    if (n_erased) *n_erased = erased;
    if (!erased) { kc_release(ctx, keep); return KC_OK; }
    u64 *nkeys = nullptr; u32 *nstate = nullptr; u8 *ntags = nullptr;
    rc = set_alloc_table(set, set->cap, &nkeys, &nstate, &ntags);
    if (rc != KC_OK) { kc_release(ctx, keep); return rc; }
    set->counts = keep;   // released by the rebuild
    KC_TRY(kcs_rebuild(set, set->cap, set->nblocks, nkeys, nstate, ntags, 1u));
    h[0] -= erased; h[1] = 0; h[2] = 0; h[3] = 0;
    KC_HIP(ctx, hipMemcpyAsync(set->counters, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    set->size_host = h[0];
    return KC_OK;
}

extern "C" int kc_set_size(kc_set *set, uint64_t *n) {
    if (!set || !n) return KC_E_ARG;
    KC_TRY(kcs_refresh_size(set));
    *n = set->size_host;
    return KC_OK;
}

extern "C" int kc_set_export(kc_set *set, uint64_t *out_words) {
    if (!set || !out_words) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    KC_TRY(kcs_refresh_size(set));
    u64 n = set->size_host;
    if (!n) return KC_OK;
    u64 *d_out = nullptr, *cursor = nullptr;
    KC_TRY(kc_alloc(ctx, &d_out, n * set->W));
    KC_TRY(kc_alloc(ctx, &cursor, 1));
    KC_HIP(ctx, hipMemsetAsync(cursor, 0, 8, ctx->stream));
    const dim3 g = kc_grid2(set->cap, 256);
    switch (set->W) {
        case 1: KC_LAUNCH(kc_export_kernel<1>, dim3(g), dim3(256), 0, ctx->stream, view_of<1>(set), d_out, cursor); break;
        case 2: KC_LAUNCH(kc_export_kernel<2>, dim3(g), dim3(256), 0, ctx->stream, view_of<2>(set), d_out, cursor); break;
        default: KC_LAUNCH(kc_export_kernel<4>, dim3(g), dim3(256), 0, ctx->stream, view_of<4>(set), d_out, cursor); break;
    }
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipMemcpyAsync(out_words, d_out, n * set->W * 8, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, d_out); kc_release(ctx, cursor);
    return KC_OK;
}

extern "C" int kc_set_contains(kc_set *set, const uint64_t *key_words, uint64_t n, uint8_t *out) {
    if (!set || (!key_words && n) || (!out && n)) return KC_E_ARG;
    kc_ctx *ctx = set->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (!n) return KC_OK;
    u64 *d_keys = nullptr; u8 *d_out = nullptr;
    KC_TRY(kc_alloc(ctx, &d_keys, n * set->W));
    KC_TRY(kc_alloc(ctx, &d_out, n));
    KC_HIP(ctx, hipMemcpyAsync(d_keys, key_words, n * set->W * 8, hipMemcpyHostToDevice, ctx->stream));
    const dim3 g = kc_grid2(n, 256);
    switch (set->W) {
        case 1: KC_LAUNCH(kc_contains_kernel<1>, dim3(g), dim3(256), 0, ctx->stream, view_of<1>(set), d_keys, n, d_out); break;
        case 2: KC_LAUNCH(kc_contains_kernel<2>, dim3(g), dim3(256), 0, ctx->stream, view_of<2>(set), d_keys, n, d_out); break;
        default: KC_LAUNCH(kc_contains_kernel<4>, dim3(g), dim3(256), 0, ctx->stream, view_of<4>(set), d_keys, n, d_out); break;
    }
    KC_HIP(ctx, hipGetLastError());
    KC_HIP(ctx, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    kc_release(ctx, d_keys); kc_release(ctx, d_out);
    return KC_OK;
}
