// Internal structures shared by the HIP translation units of libkmercamel_amd.so. Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/kmercamel_amd.h"
#include "kc_key.h"
#include "kc_set_dev.h"

typedef unsigned long long ull;

struct KcPool {
    struct Blk { size_t size; void *p; };
    std::vector<Blk> free_list;
    std::map<void *, size_t> live;
    size_t live_bytes = 0, peak_bytes = 0, malloc_bytes = 0, malloc_calls = 0;
    double malloc_seconds = 0;   // wall time inside hipMalloc (large first-time allocations dominate a cold CLI run)
    hipError_t alloc(void **out, size_t bytes);
    void release(void *p);
    void trim();      // hipFree everything cached
    ~KcPool();
};

struct kc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool err_agreed = false;   // the last error of a collective call was one that every rank takes at the same point (kc_fail_agreed; cleared on entry)
    KcPool pool;
    hipEvent_t ev[8] = {nullptr};
    // dominant-kernel timing (filled by the pipeline)
    double dom_us = 0; uint64_t dom_launches = 0;
    // streamed CLI I/O (kc_io.hip): a second stream so that the host <-> device copy of chunk i+1 runs under the kernels of chunk i, and
    // the pinned chunks themselves, kept for the life of the context (measured on the MI355X box: hipHostMalloc 0.15 ms per MB,
    // hipHostFree 0.11 ms per MB -- the input's chunks serve the output again)
    hipStream_t copy_stream = nullptr;
    struct Pin { void *p; size_t bytes; bool busy; };
    std::vector<Pin> pinned;
};
void *kc_pin_acquire(kc_ctx *ctx, size_t bytes);   // kc_io.hip; nullptr on failure
void kc_pin_release(kc_ctx *ctx, void *p);
void kc_pin_free_all(kc_ctx *ctx);

struct kc_set {
    kc_ctx *ctx = nullptr;
    int k = 0, W = 1, complements = 1;
    u64 cap = 0;          // slots = nblocks << bshift
    u64 nblocks = 0;
    int bshift = 12;      // 4096 slots per block
    int mlen = 0;         // minimizer length (0 = plain hashing)
    u32 local_probes = 128;
    u64 *keys = nullptr;  // cap * W words; W==1: EMPTY = ~0
    u32 *state = nullptr; // W>1: 0 empty, 1 busy, 2 full
    u8 *tags = nullptr;   // cap bytes
    u32 *counts = nullptr; // cap entries, only while counting occurrences for -z
    u64 *counters = nullptr;  // device: [0] size, [1] overflow flag
    u64 size_host = 0;
    u64 rlo = 0, rcap = 0;   // hash-sharded sets: the global slots [rlo, rlo + rcap) live here (rcap == 0: all of them); keys / state / tags / counts
                             // are then LOCAL arrays of rcap entries
    u32 *ncount = nullptr;   // dense node ids: k-mers per block (blocks of this rank), see kcs_prepare_dense and SetView
    u32 nstride = 0;         // ids per block = the largest ncount, rounded up to a multiple of 8
    float occ_ratio = 0;     // occurrences per distinct k-mer in the sample the bucketed build took of a fresh set (0: not sampled)
    bool dense_ok = false;   // ncount / nstride and the counts in the empty slots describe the table as it is (any insertion voids them)
    bool homes_primary = false;   // every key sits in the local probe run of its PRIMARY home (a fresh bucketed build without spills, nothing inserted
                                  // since): what the probe kernel's in-block guess (kc_probe_block_kernel<..., GP>) relies on
    mutable bool unclean = false;   // table allocated but not filled with EMPTY yet (a bucketed build of a fresh set overwrites every
                                    // slot, so the 9 B/slot fill is deferred until something else needs the table)
};
int kcs_clean_table(const kc_set *s);   // kc_set.hip

template <int W> static inline SetView<W> kc_view_raw(const kc_set *s) {
    const u64 rcap = s->rcap ? s->rcap : s->cap;
    return SetView<W>{s->keys - s->rlo * W, s->state ? s->state - s->rlo : nullptr, s->tags - s->rlo, s->cap, s->nblocks, s->bshift, s->mlen, s->k,
                      s->local_probes, s->rlo, rcap, s->complements, s->ncount ? s->ncount - (s->rlo >> s->bshift) : nullptr, s->nstride};
}
// every kernel reaches the table through a view: this is where a deferred fill happens
template <int W> static inline SetView<W> kc_view(const kc_set *s) {
    // (a failed fill leaves its message in the context and keeps the flag set: the next call that checks HIP errors reports it)
    if (s->unclean && kcs_clean_table(s) != KC_OK) s->unclean = true;
    return kc_view_raw<W>(s);
}

struct kc_simplitigs {
    kc_ctx *ctx = nullptr;
    int k = 0, W = 1;
    u64 n = 0, total_bases = 0, nwords = 0;
    u64 *lens = nullptr;   // n: bases per simplitig
    u64 *woff = nullptr;   // n+1: first 64-bit word of each simplitig (32 bases per word, MSB first, word aligned)
    u64 *words = nullptr;  // nwords
    u64 *firstK = nullptr; // n*W: first k-mer
    u64 *lastK = nullptr;  // n*W: last k-mer
    float occ_ratio = 0;   // occurrences per distinct k-mer of the set they were built from (kc_set::occ_ratio; 0: unknown) -- a hint for phase 2
};

struct kc_result {
    kc_ctx *ctx = nullptr;
    char *d_out = nullptr, *d_mask = nullptr;   // the slice [slice_off, slice_off + slice_len) of the superstring (single GPU: all of it)
    u64 length = 0;                              // length of the whole superstring
    u64 slice_off = 0, slice_len = 0;
    u64 stats[16] = {0};
    u64 stats_ext[16] = {0};   // [0] cycles cut by the simplitig rule, [1] k-mers on this rank, [2] neighbour queries sent (sharded runs)
};

// Timing experiments that SKIP work (stores, inserts, loads) and therefore produce WRONG results exist only in the tools' build
// (`make experiments` -> libkmercamel_amd_exp.so, -DKC_EXPERIMENTS): in the shipped library the macro is a null pointer and the
// variable's name is not even in the binary (tests/test_cpu_oracle.py checks `strings`).
#ifdef KC_EXPERIMENTS
#define KC_EXP_KNOB(name) getenv(name)
#else
#define KC_EXP_KNOB(name) ((const char *)nullptr)
#endif

// Set by kc_grid when a launch would need 2^32 or more threads in one dimension (a dispatch carries its size as 32 bits per
// dimension).  KC_LAUNCH -- the only way this library launches a kernel -- tests it right after the grid expression has been
// evaluated and returns KC_E_ARG from the calling function BEFORE anything is launched, so no later kernel ever sees the output a
// skipped launch would have left undefined.  Both KC_LAUNCH and KC_HIP consume the flag on every use, so it cannot go stale.
inline thread_local bool kc_grid_overflowed = false;
static inline bool kc_take_grid_overflow() { const bool b = kc_grid_overflowed; kc_grid_overflowed = false; return b; }
#define KC_GRID_MSG "the input needs a 1-D launch of 2^32 or more threads (not supported)"
#define KC_HIP(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        const bool big__ = kc_take_grid_overflow();                                                        \
        if (e__ != hipSuccess || big__) {                                                                  \
            (ctx)->err = std::string(big__ ? KC_GRID_MSG ": " : "") + #call + ": " + hipGetErrorString(e__) + " @" + __FILE__ + ":" + std::to_string(__LINE__); \
            return big__ ? KC_E_ARG : KC_E_NO_DEVICE;                                                      \
        }                                                                                                  \
    } while (0)
// (needs `kc_ctx *ctx` in scope and a calling function that returns a KC_* status)
#define KC_LAUNCH(kern, grid, block, shm, strm, ...)                                                        \
    do {                                                                                                   \
        const dim3 grid__ = (grid);                                                                        \
        if (kc_take_grid_overflow()) {                                                                     \
            ctx->err = std::string(KC_GRID_MSG) + " @" + __FILE__ + ":" + std::to_string(__LINE__);         \
            return KC_E_ARG;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL(kern, grid__, block, shm, strm, __VA_ARGS__);                                    \
    } while (0)

#define KC_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != KC_OK) return rc__; \
    } while (0)

static inline int kc_fail(kc_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// An error of the hash-sharded flow that all ranks return together (same test on replicated or all-reduced data): the peers are not
// left waiting in an exchange, so kcc_after_collective_call must not tear the transport down.
static inline int kc_fail_agreed(kc_ctx *ctx, int code, const std::string &msg) {
    if (ctx) { ctx->err = msg; ctx->err_agreed = true; }
    return code;
}

template <typename T> static inline int kc_alloc(kc_ctx *ctx, T **out, size_t count) {
    void *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t e = ctx->pool.alloc(&p, bytes);
    if (e != hipSuccess) {
        ctx->err = std::string("device allocation of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e);
        return KC_E_NO_DEVICE;
    }
    *out = (T *)p;
    return KC_OK;
}
static inline void kc_release(kc_ctx *ctx, void *p) { if (p) ctx->pool.release(p); }

// Temporaries of one call: everything allocated through the scope goes back to the pool when the scope ends, whatever
// the exit path (KcPool::release ignores pointers it does not hold, so an explicit earlier kc_release stays harmless).
struct KcScope {
    kc_ctx *ctx;
    std::vector<void *> ptrs;
    explicit KcScope(kc_ctx *c) : ctx(c) {}
    KcScope(const KcScope &) = delete;
    KcScope &operator=(const KcScope &) = delete;
    ~KcScope() {
        if (!ptrs.empty()) (void)hipStreamSynchronize(ctx->stream);   // nothing in flight may still use them
        for (void *p : ptrs) kc_release(ctx, p);
    }
    template <typename T> int alloc(T **out, size_t count) {
        int rc = kc_alloc(ctx, out, count);
        if (rc == KC_OK) ptrs.push_back((void *)*out);
        return rc;
    }
    void keep(void *p) { for (auto &q : ptrs) if (q == p) q = nullptr; }   // ownership moves to the caller
    void release(void *p) {                                                // early release of one temporary
        if (!p) return;
        for (auto &q : ptrs) if (q == p) q = nullptr;   // forget it first: the pool may hand the block to someone else
        kc_release(ctx, p);
    }
};

// A dispatch carries its size in work-items as 32 bits per dimension: a 1-D launch of 2^32 or more threads is silently
// truncated.  kc_grid refuses such a launch (the call fails with KC_E_ARG); kernels that sweep table-sized ranges (> 2^32 slots at 2.5 G k-mers)
// are launched on kc_grid2's 2-D grid and number their threads with kc_gtid().
// (test knob, read once: KC_TEST_GRID_LIMIT=n makes launches of n or more threads count as overflowing, so that the refusal can be
// driven through a real call without 2^32 work items)
static inline bool kc_grid_test_limit_hit(u64 threads) {
    static const u64 lim = [] { const char *e = getenv("KC_TEST_GRID_LIMIT"); return e ? strtoull(e, nullptr, 10) : 0ull; }();
    return lim && threads >= lim;
}
static inline unsigned kc_grid(u64 n, unsigned block) {
    u64 g = (n + block - 1) / block;
    if (g * block >= ((u64)1 << 32) || kc_grid_test_limit_hit(g * block)) { kc_grid_overflowed = true; return 1u; }   // (KC_LAUNCH refuses the launch)
    return (unsigned)(g ? g : 1);
}
static inline dim3 kc_grid2(u64 n, unsigned block) {
    u64 g = (n + block - 1) / block;
    if (!g) g = 1;
    const unsigned gx = g > ((u64)1 << 22) ? (1u << 22) : (unsigned)g;
    return dim3(gx, (unsigned)((g + gx - 1) / gx));
}
#if defined(__HIPCC__)
__device__ __forceinline__ u64 kc_gtid() { return ((u64)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; }
#endif
// run something on every exit path of a scope (early error returns included)
template <typename F> struct KcDefer { F f; ~KcDefer() { f(); } };
template <typename F> static inline KcDefer<F> kc_defer(F f) { return KcDefer<F>{f}; }

static inline unsigned kc_grid8(u64 n, unsigned block) { u64 g = (n + block - 1) / block; g = (g + 7) / 8 * 8; return (unsigned)(g ? g : 8); }  // for kc_xcd_block_id()

// ---- kc_util.hip: scan / ordered compaction / sharded lists / message routing (hand-written, no library calls) ----
int kcu_exclusive_scan(kc_ctx *ctx, const u64 *in, u64 *out, u64 n, bool sync = true);   // out[0..n], out[n] = total; out may alias in
int kcu_select_flagged(kc_ctx *ctx, const u64 *in, const u8 *flags, u64 n, u64 *out, u64 *d_total, bool sync = true);   // stable
int kcp_sort_perm_by_keys(kc_ctx *ctx, const u64 *d_keys, u64 n, int W, int k, u64 **perm_out);   // kc_simplitigs.hip: *perm_out (kc_alloc'd) = the rows of d_keys (n x W words) in ascending key order
int kcu_sort_rows_bucketed(kc_ctx *ctx, const u64 *keys, const u64 *ids, u64 n, int W, int total_bits, u64 *keys_out, u64 *ids_out, u64 *d_flag);   // MSD bucket pass + LDS sorts; KC_E_ARG: not applicable
int kcu_sort_pairs(kc_ctx *ctx, const u64 *keys_in, u64 *keys_out, const u64 *vals_in, u64 *vals_out, u64 n, int begin_bit, int end_bit, bool sync = true);   // stable LSD radix sort
int kcu_compact_shards(kc_ctx *ctx, const u64 *src, u64 shard_cap, const u64 *d_cursors, int nshards, int stride, int E, u64 *dst, u64 *total, bool *overflow);
struct kc_comm;
struct KcRoute {
    u64 n = 0, n_recv = 0;
    u64 n_total = 0;                          // items of all ranks together (every rank knows the whole count matrix)
    u64 *perm = nullptr;                      // device: send position -> item
    std::vector<u64> scnt, soff, rcnt, roff;  // items per peer
};
int kcr_build(kc_comm *c, const u8 *dest, u64 n, KcRoute *r);
void kcr_free(kc_ctx *ctx, KcRoute *r);
int kcr_forward(kc_comm *c, const KcRoute &r, const u64 *items, int E, int stride, u64 **recv_out);
int kcr_backward(kc_comm *c, const KcRoute &r, const u64 *answers, int A, u64 *out);

// ---- cross-TU host entry points (each dispatches on W internally) ----
// kc_set.hip
int kcs_pack(kc_ctx *ctx, const void *d_seq, u64 len, u64 **d_codes, u64 **d_brk, u64 **d_low);
int kcs_reserve(kc_set *set, u64 additional);
int kcs_insert_packed(kc_set *set, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int case_sensitive);
int kcs_refresh_size(kc_set *set);
int kcs_build_counted(kc_ctx *ctx, int k, int complements, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, int min_count, kc_set **out);   // -z through the LDS counting build
struct kc_comm;
int kcs_prepare_dense_sharded(kc_set *set, kc_comm *c);   // the same on a hash-sharded set + one nstride for all ranks (collective)
int kcs_prepare_dense(kc_set *set);   // dense node ids for phase 1: counts in the empty slots, set->ncount / nstride (no-op when still valid)
int kcs_create_sharded(kc_ctx *ctx, kc_comm *c, int k, int complements, u64 occ_local, kc_set **out);   // collective
int kcs_build_sharded(kc_set *set, kc_comm *c, const u64 *d_codes, const u64 *d_brk, const u64 *d_low, u64 len, u64 *moved_bytes, int min_count = 1);   // collective; min_count > 1: -z
// kc_simplitigs.hip
int kcp_build_simplitigs(kc_set *set, kc_simplitigs **out, u64 *rounds, u64 *cycles_cut);
int kcp_simplitigs_from_device_ascii(kc_ctx *ctx, int k, const u8 *d_seq, u64 len, const u64 *h_offsets, u64 n, int offsets_are_host_positions,
                                     kc_simplitigs **out, int gap = 0);
int kcp_record_starts_from_separators(kc_ctx *ctx, const u8 *d_seq, u64 len, std::vector<u64> &offs);
int kcp_to_sparse(kc_simplitigs *s, kc_simplitigs **out);
int kcp_insert_simplitig_kmers(kc_set *set, kc_simplitigs *s);
// kc_overlap.hip
struct KcPathDev { u64 V = 0; u64 *edge_from = nullptr; u8 *overlaps = nullptr; u64 *pred = nullptr; u64 repairs = 0; };
struct kc_comm;
int kco_overlap_path_device(kc_simplitigs *s, int complements, int lower_bound, KcPathDev *out, kc_comm *comm = nullptr);   // comm: keys of a level divided among the ranks
void kco_path_release(kc_ctx *ctx, KcPathDev *p);
// kc_emit.hip
struct KcRank { u64 V = 0; u64 *dist = nullptr, *endv = nullptr, *weight = nullptr; u64 start = 0, L = 0; };
int kce_rank_path(kc_simplitigs *s, const KcPathDev *path, KcRank *r);
void kce_rank_release(kc_ctx *ctx, KcRank *r);
int kce_emit_ranked(kc_simplitigs *s, const KcRank *r, u64 out_lo, int nranks, int me, char *out);
int kce_superstring(kc_simplitigs *s, const KcPathDev *path, int complements, kc_set *set_for_mask, char **d_out, char **d_mask, u64 *length);
int kce_maskopt(kc_ctx *ctx, const char *ms, u64 len, int k, int complements, int minimize, char *out);
