// C-ABI glue: Global / SuperstringFromPath entry points and the fused, device-resident greedy branch of
// kmercamel<>() (main.cpp:122-212).
#include "kc_internal.h"
#include "kc_comm.h"

static int copy_out_string(kc_ctx *ctx, const char *d, u64 L, char **out) {
    char *h = (char *)malloc(L + 1);
    if (!h) return kc_fail(ctx, KC_E_ARG, "host allocation failed");
    if (L) {
        if (hipMemcpyAsync(h, d, L, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            free(h);
            return kc_fail(ctx, KC_E_NO_DEVICE, "D2H copy failed");
        }
    }
    h[L] = 0;
    *out = h;
    return KC_OK;
}

extern "C" int kc_superstring_from_path(kc_simplitigs *s, const uint64_t *edge_from, const uint8_t *overlaps, int complements,
                                        kc_set *set, char **out, char **mask_out, uint64_t *length) {
    if (!s || !edge_from || !overlaps || !out) return KC_E_ARG;
    kc_ctx *ctx = s->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (s->n == 0) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
    KcPathDev p;
    char *d_out = nullptr, *d_mask = nullptr;
    u64 L = 0;
    auto cleanup = kc_defer([&] { (void)hipStreamSynchronize(ctx->stream); kc_release(ctx, d_out); kc_release(ctx, d_mask); kco_path_release(ctx, &p); });
    p.V = s->n * (complements ? 2 : 1);
    KC_TRY(kc_alloc(ctx, &p.edge_from, p.V));
    KC_TRY(kc_alloc(ctx, &p.overlaps, p.V));
    KC_HIP(ctx, hipMemcpyAsync(p.edge_from, edge_from, p.V * 8, hipMemcpyHostToDevice, ctx->stream));
    KC_HIP(ctx, hipMemcpyAsync(p.overlaps, overlaps, p.V, hipMemcpyHostToDevice, ctx->stream));
    int rc = kce_superstring(s, &p, complements ? 1 : 0, mask_out ? set : nullptr, &d_out, mask_out ? &d_mask : nullptr, &L);
    if (rc == KC_OK) rc = copy_out_string(ctx, d_out, L, out);
    if (rc == KC_OK && mask_out) {
        if (d_mask) rc = copy_out_string(ctx, d_mask, L, mask_out);
        else *mask_out = nullptr;
    }
    if (length) *length = L;
    return rc;
}

extern "C" int kc_global(kc_simplitigs *s, int complements, kc_set *set, int want_mask, char **out, char **mask_out, uint64_t *length) {
    if (!s || !out) return KC_E_ARG;
    kc_ctx *ctx = s->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (s->n == 0) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
    kc_set *own = nullptr;
    if (want_mask && mask_out && !set) {   // fill_kmers (simplitigs.h:41-51)
        KC_TRY(kc_set_create(ctx, s->k, complements ? 1 : 0, s->total_bases, &own));
        int rc = kcp_insert_simplitig_kmers(own, s);
        if (rc != KC_OK) { kc_set_destroy(own); return rc; }
        set = own;
    }
    KcPathDev p;
    int rc = kco_overlap_path_device(s, complements ? 1 : 0, 0, &p);
    char *d_out = nullptr, *d_mask = nullptr;
    u64 L = 0;
    bool mask = want_mask && mask_out && set;
    if (rc == KC_OK) rc = kce_superstring(s, &p, complements ? 1 : 0, mask ? set : nullptr, &d_out, mask ? &d_mask : nullptr, &L);
    if (rc == KC_OK) rc = copy_out_string(ctx, d_out, L, out);
    if (rc == KC_OK && mask) rc = copy_out_string(ctx, d_mask, L, mask_out);
    if (length) *length = L;
    kc_release(ctx, d_out); kc_release(ctx, d_mask);
    kco_path_release(ctx, &p);
    if (own) kc_set_destroy(own);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------
static float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ms;
}

extern "C" int kc_compute(kc_ctx *ctx, const void *seq, uint64_t len, const uint64_t *offsets, uint64_t n_records, int k, uint32_t flags,
                          kc_result **out) {
    if (!ctx || !out || (!seq && len)) return KC_E_ARG;
    *out = nullptr;
    if (k < 1 || k > 127) return kc_fail(ctx, KC_E_ARG, "k must be in 1..127");
    KC_HIP(ctx, hipSetDevice(ctx->device));
    const int complements = (flags & KC_FLAG_COMPLEMENTS) ? 1 : 0;
    const bool from_simplitigs = (flags & KC_FLAG_SIMPLITIGS_INPUT) != 0;
    const bool want_mask = (flags & KC_FLAG_WANT_MASK) != 0;
    const bool on_device = (flags & KC_FLAG_INPUT_ON_DEVICE) != 0;
    const int min_frequency = (int)((flags >> 8) & 0xff);
    if (min_frequency > 1 && from_simplitigs) return kc_fail(ctx, KC_E_ARG, "Inputting simplitigs is not compatible with frequency filterring.");
    if (from_simplitigs && !offsets && !on_device) return kc_fail(ctx, KC_E_ARG, "-S input needs record offsets");
    // (-S with the sequence on the device and no offsets: the records are separated by '\n', as kc_read_fasta_device leaves them)
    std::vector<u64> sep_offs;
    int gap = 0;
    if (from_simplitigs && !offsets) {
        KC_TRY(kcp_record_starts_from_separators(ctx, (const u8 *)seq, len, sep_offs));
        if (sep_offs.size() < 2) return kc_fail(ctx, KC_E_EMPTY, "input cannot be empty");
        offsets = sep_offs.data(); n_records = sep_offs.size() - 1; gap = 1;
    }

    hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1], e2 = ctx->ev[2], e3 = ctx->ev[3];
    hipEvent_t eend;
    KC_HIP(ctx, hipEventCreate(&eend));
    u8 *d_own = nullptr;
    kc_result *r = nullptr;
    kc_set *set = nullptr;
    kc_simplitigs *simp = nullptr, *sparse = nullptr;
    bool done = false;
    KcPathDev p;
    // whatever the exit path: the event, the intermediate objects and (unless the call succeeds) the result go back
    auto cleanup = kc_defer([&] {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipEventDestroy(eend);
        kco_path_release(ctx, &p);
        if (sparse) kc_simplitigs_destroy(sparse);
        if (simp) kc_simplitigs_destroy(simp);
        if (set) kc_set_destroy(set);
        kc_release(ctx, d_own);
        if (!done && r) kc_result_destroy(r);
    });
    const u8 *d_seq = (const u8 *)seq;
    if (!on_device) {
        KC_TRY(kc_alloc(ctx, &d_own, len + 64));
        KC_HIP(ctx, hipMemcpyAsync(d_own, seq, len, hipMemcpyHostToDevice, ctx->stream));
        d_seq = d_own;
    }
    r = new kc_result();
    r->ctx = ctx;
    u64 rounds = 0, n_kmers = 0, cycles = 0;
    int rc = KC_OK;
    double dom_us = 0;
    KC_HIP(ctx, hipEventRecord(e0, ctx->stream));
    if (!from_simplitigs) {
        // phase 0: ReadKMers/AddKMers (parser.h:22-49,106-118)
        u64 occ = len >= (u64)k ? len - k + 1 : 0;
        u64 *codes = nullptr, *brk = nullptr, *low = nullptr;
        // -z (ReadKMersFiltered, parser.h:122-141) counts occurrences.  The direct insert does that with one global compare-and-swap and one
        // global counter update per OCCURRENCE (2.3 G/s measured on 10x reads: 213 ms for 5 10^8 occurrences, 24 times the bucketed build
        // without -z); the hash-sharded flow's build counts where a block's occurrences meet, in LDS (kcs_build_counted).  What that does
        // not cover (k < 18, blocks that overflow) falls through to the direct insert.  KC_Z_DIRECT=1: the direct insert only.
        bool counted = false;
        if (min_frequency > 1 && k >= 18 && len >= (u64)k && !getenv("KC_Z_DIRECT")) {
            rc = kcs_pack(ctx, d_seq, len, &codes, &brk, &low);
            if (rc == KC_OK) {
                (void)hipEventRecord(ctx->ev[6], ctx->stream);
                rc = kcs_build_counted(ctx, k, complements, codes, brk, low, len, min_frequency, &set);
                (void)hipEventRecord(ctx->ev[7], ctx->stream);
                if (rc == KC_OK) counted = true;
                else if (ctx->err_agreed && (rc == KC_E_UNSUPPORTED_SHARDED || rc == KC_E_ARG || rc == KC_E_CAPACITY)) { rc = KC_OK; set = nullptr; }   // (the direct insert takes it)
            }
            if (!counted) { kc_release(ctx, codes); kc_release(ctx, brk); kc_release(ctx, low); codes = brk = low = nullptr; }
        }
        if (rc == KC_OK && !counted) rc = kc_set_create(ctx, k, complements, occ, &set);
        if (rc == KC_OK && !counted && min_frequency > 1) rc = kc_set_enable_counting(set);
        if (rc == KC_OK && !counted && len >= (u64)k) {
            // the port links of phase 1 are the largest block of the run and the build buckets of phase 0 (8 W B per slot) are gone
            // by then: obtain the larger of the two now, so that both phases use the same memory.  The links are 16 B per node id,
            // nstride ids per block (the largest block: ~0.64 of a block's slots on uniform data at load 0.5): 0.7 x 16 B per slot is
            // reserved; a table with a fuller block simply obtains a larger block of its own when phase 1 starts.
            // (Beyond ~3.2 10^9 k-mers the table is built at load 0.66 and its fullest block holds 0.86 of a block's slots -- 7 068 of 8 192 at
            // 4.4 10^9 k-mers: 0.9 x 16 B per slot then.  With 0.7 there the links missed the reserved block by 18 GB, the pool held 77 GB it
            // could not use, ran out of device memory, trimmed and allocated again behind the driver's scrub: 2.0 - 2.4 s inside hipMalloc
            // of a fresh process, 269 GB allocated for a peak of 182.)
            void *arena = nullptr;
            const bool dense_load = (double)occ > 0.55 * (double)set->cap;
            const size_t arena_bytes = set->W == 1 ? (size_t)(set->cap / 4 * (dense_load ? 58 : 45)) : (size_t)set->cap * 16;
            if (ctx->pool.alloc(&arena, arena_bytes) == hipSuccess) ctx->pool.release(arena); else (void)hipGetLastError();
            rc = kcs_pack(ctx, d_seq, len, &codes, &brk, &low);
            if (rc == KC_OK) rc = kcs_insert_packed(set, codes, brk, low, len, 0);
        }
        if (rc == KC_OK) rc = kcs_refresh_size(set);
        if (rc == KC_OK && min_frequency > 1 && !counted) rc = kc_set_filter_min_frequency(set, min_frequency);   // ReadKMersFiltered (parser.h:122-141)
        kc_release(ctx, codes); kc_release(ctx, brk); kc_release(ctx, low);
        if (rc == KC_OK && set->size_host == 0) rc = kc_fail(ctx, KC_E_NO_KMERS, "input contains no k-mers");
        if (rc == KC_OK) { n_kmers = set->size_host; dom_us += 1000.0 * ev_ms(ctx->ev[6], ctx->ev[7]); }
        KC_HIP(ctx, hipEventRecord(e1, ctx->stream));
        // phase 1: get_simplitigs (simplitigs.h:191-205)
        if (rc == KC_OK) rc = kcp_build_simplitigs(set, &simp, &rounds, &cycles);
        if (rc == KC_OK) r->stats[10] = (u64)(1000.0 * ev_ms(ctx->ev[4], ctx->ev[5]));   // neighbour-probe kernel
    } else {
        KC_HIP(ctx, hipEventRecord(e1, ctx->stream));
        // simplitigs_from_fasta (simplitigs.h:89-103)
        rc = kcp_simplitigs_from_device_ascii(ctx, k, d_seq, len, offsets, n_records, 1, &simp, gap);
        if (rc == KC_OK) n_kmers = simp->total_bases - simp->n * (u64)(k - 1);
        if (rc == KC_OK && want_mask) {
            rc = kc_set_create(ctx, k, complements, simp->total_bases, &set);
            if (rc == KC_OK) rc = kcp_insert_simplitig_kmers(set, simp);
        }
    }
    KC_HIP(ctx, hipEventRecord(e2, ctx->stream));
    u64 S = 0;
    kc_simplitigs *use = simp;
    if (rc == KC_OK) {
        S = simp->n;
        // main.cpp:175: too many simplitigs => overlap graph over single k-mers (global_sparse.h)
        if (!from_simplitigs && S * 5 >= n_kmers) {
            rc = kcp_to_sparse(simp, &sparse);
            if (rc == KC_OK) { use = sparse; r->stats[3] = 1; }
        }
    }
    if (rc == KC_OK) rc = kco_overlap_path_device(use, complements, 0, &p);
    KC_HIP(ctx, hipEventRecord(e3, ctx->stream));
    if (rc == KC_OK) {
        rc = kce_superstring(use, &p, complements, want_mask ? set : nullptr, &r->d_out, want_mask ? &r->d_mask : nullptr, &r->length);
        r->stats[4] = p.V; r->stats[13] = p.repairs;
        kco_path_release(ctx, &p);
    }
    KC_HIP(ctx, hipEventRecord(eend, ctx->stream));
    KC_HIP(ctx, hipEventSynchronize(eend));
    r->slice_off = 0; r->slice_len = r->length;
    if (rc == KC_OK) {
        r->stats[0] = n_kmers; r->stats[1] = S; r->stats[2] = r->length; r->stats[5] = rounds;
        r->stats[6] = (u64)(1000.0 * ev_ms(e0, e1)); r->stats[7] = (u64)(1000.0 * ev_ms(e1, e2));
        r->stats[8] = (u64)(1000.0 * ev_ms(e2, e3)); r->stats[9] = (u64)(1000.0 * ev_ms(e3, eend));
        r->stats[11] = (u64)dom_us;   // insert kernel
        r->stats[12] = set ? set->cap : 0;
        r->stats[14] = (u64)(1000.0 * ev_ms(e0, eend));
        r->stats_ext[0] = cycles; r->stats_ext[1] = n_kmers;
    }
    if (rc != KC_OK) return rc;
    done = true;
    *out = r;
    return KC_OK;
}

extern "C" int kc_result_stats(kc_result *r, uint64_t *stats) {
    if (!r || !stats) return KC_E_ARG;
    memcpy(stats, r->stats, sizeof r->stats);
    return KC_OK;
}
extern "C" int kc_result_stats_ext(kc_result *r, uint64_t *stats, int n) {
    if (!r || !stats || n < 0) return KC_E_ARG;
    for (int i = 0; i < n; ++i) stats[i] = i < 16 ? r->stats[i] : (i < 32 ? r->stats_ext[i - 16] : 0);
    return KC_OK;
}
extern "C" int kc_result_device_ptrs(kc_result *r, void **d_out, void **d_mask) {
    if (!r) return KC_E_ARG;
    if (d_out) *d_out = r->d_out;
    if (d_mask) *d_mask = r->d_mask;
    return KC_OK;
}
extern "C" int kc_result_copy(kc_result *r, char *out, char *mask_out) {
    if (!r || !out) return KC_E_ARG;
    kc_ctx *ctx = r->ctx;
    KC_HIP(ctx, hipSetDevice(ctx->device));
    if (r->slice_len) KC_HIP(ctx, hipMemcpyAsync(out, r->d_out, r->slice_len, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out && r->d_mask && r->slice_len) KC_HIP(ctx, hipMemcpyAsync(mask_out, r->d_mask, r->slice_len, hipMemcpyDeviceToHost, ctx->stream));
    KC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KC_OK;
}
extern "C" int kc_result_slice(kc_result *r, uint64_t *offset, uint64_t *slice_len) {
    if (!r) return KC_E_ARG;
    if (offset) *offset = r->slice_off;
    if (slice_len) *slice_len = r->slice_len;
    return KC_OK;
}
extern "C" void kc_result_destroy(kc_result *r) {
    if (!r) return;
    (void)hipStreamSynchronize(r->ctx->stream);
    kc_release(r->ctx, r->d_out); kc_release(r->ctx, r->d_mask);
    delete r;
}
