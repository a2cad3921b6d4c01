// Drop-in for the reference's src/khash_utils.h on the hot path: the k-mer set types kh_S64_t/kh_S128_t/kh_S256_t and
// the by-value wrapper tags kmer_dict64_t/128_t/256_t (khash_utils.h:28-95), backed by the device-resident set of
// libkmercamel_amd.so instead of khash.  Only what the greedy path and its callers use is provided:
// kh_init_set / kh_destroy_set / kh_put_to_set / kh_get_from_set / kh_del_from_set, kh_size, kh_end, containsKMer, eraseKMer,
// nextKMer, kMersToVec, kMersFromVec.
// Element-wise calls (kh_put_to_set, kh_get_from_set) cross the ABI once per call -- they exist for compatibility; bulk
// ingestion goes through AddKMers / ReadKMers (parser.h) which hand whole records to one kernel.
#pragma once
#include <vector>
#include <algorithm>
#include "kmers.h"
#include "kc_runtime.h"

typedef uint64_t khint_t;

template <typename kmer_t> struct kc_hset {
    kc_set *h = nullptr;
    int k = 0;            // fixed at the first AddKMers / put (the reference's sets are k-agnostic integers)
    bool complements = true;
    uint64_t size = 0;
    std::vector<kmer_t> pending;   // keys put one by one before k is known
    kmer_t last_get = kmer_t(0);   // key of the last successful kh_get_from_set (what a following kh_del_from_set removes)
    std::vector<kmer_t> order;     // nextKMer: the keys in the order the set was first iterated
    bool have_order = false;
};
typedef kc_hset<kmer64_t> kh_S64_t;
typedef kc_hset<kmer128_t> kh_S128_t;
typedef kc_hset<kmer256_t> kh_S256_t;

#define kh_size(h) ((h)->size)
#define kh_end(h) ((khint_t)~0ull)

namespace kcamd {
template <typename kmer_t> inline int width_k(); // smallest k that selects this key width in the ABI
template <> inline int width_k<kmer64_t>() { return 31; }
template <> inline int width_k<kmer128_t>() { return 63; }
template <> inline int width_k<kmer256_t>() { return 127; }

template <typename kmer_t> void ensure(kc_hset<kmer_t> *s, int k, bool complements) {
    if (s->h) return;
    s->k = k; s->complements = complements;
    check(kc_set_create(context(), k, complements ? 1 : 0, 0, &s->h));
}
template <typename kmer_t> void refresh(kc_hset<kmer_t> *s) { check(kc_set_size(s->h, &s->size)); }
}  // namespace kcamd

#define KC_INIT_WRAPPER(type)                                                                                              \
    struct kmer_dict##type##_t {                                                                                           \
        inline kh_S##type##_t *kh_init_set() { return new kh_S##type##_t(); }                                              \
        inline void kh_destroy_set(kh_S##type##_t *set) { if (set) { if (set->h) kc_set_destroy(set->h); delete set; } }   \
        /* exact-key put: keys are stored as given (callers pass canonical k-mers), k = width maximum */                   \
        inline khint_t kh_put_to_set(kh_S##type##_t *set, kmer##type##_t key, int *ret) {                                  \
            kcamd::ensure(set, kcamd::width_k<kmer##type##_t>(), false);                                                    \
            uint64_t w[4] = {0, 0, 0, 0};                                                                                   \
            kcamd::to_words(key, w);                                                                                        \
            uint8_t present = 0;                                                                                            \
            kcamd::check(kc_set_contains(set->h, w, 1, &present));                                                          \
            if (!present) {                                                                                                 \
                std::string s = NumberToKMer(key, set->k);                                                                  \
                kcamd::check(kc_set_add_sequence(set->h, s.data(), s.size(), 0));                                           \
                kcamd::refresh(set);                                                                                        \
            }                                                                                                               \
            if (ret) *ret = present ? 0 : 1;                                                                                \
            return 0;                                                                                                       \
        }                                                                                                                   \
        inline khint_t kh_get_from_set(kh_S##type##_t *set, kmer##type##_t key) {                                          \
            if (!set->h) return kh_end(set);                                                                                \
            uint64_t w[4] = {0, 0, 0, 0};                                                                                   \
            kcamd::to_words(key, w);                                                                                        \
            uint8_t present = 0;                                                                                            \
            kcamd::check(kc_set_contains(set->h, w, 1, &present));                                                          \
            if (present) set->last_get = key;                                                                               \
            return present ? 0 : kh_end(set);                                                                               \
        }                                                                                                                   \
        /* kh_del (khash.h:353-359): `it` is what kh_get_from_set just returned for the key to remove */                   \
        inline void kh_del_from_set(kh_S##type##_t *set, khint_t it) {                                                      \
            if (!set->h || it == kh_end(set)) return;                                                                       \
            uint64_t w[4] = {0, 0, 0, 0};                                                                                   \
            kcamd::to_words(set->last_get, w);                                                                              \
            kcamd::check(kc_set_erase(set->h, w, 1, nullptr));                                                              \
            kcamd::refresh(set);                                                                                            \
        }                                                                                                                   \
    };
KC_INIT_WRAPPER(64)
KC_INIT_WRAPPER(128)
KC_INIT_WRAPPER(256)

/// Determine whether the k-mer or its reverse complement is present (khash_utils.h:98-103).
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
inline bool containsKMer(kh_S_t *kMers, kh_wrapper_t wrapper, kmer_t kMer, int k, bool complements) {
    bool ret = wrapper.kh_get_from_set(kMers, kMer) != kh_end(kMers);
    if (complements) ret |= wrapper.kh_get_from_set(kMers, ReverseComplement(kMer, k)) != kh_end(kMers);
    return ret;
}

/// Remove the k-mer and its reverse complement (khash_utils.h:105-118).
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
void eraseKMer(kh_S_t *kMers, kh_wrapper_t, kmer_t kMer, int k, bool complements) {
    if (!kMers->h) return;
    uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int W = kcamd::words_of<kmer_t>::value;
    kcamd::to_words(kMer, w);
    if (complements) kcamd::to_words(ReverseComplement(kMer, k), w + W);
    kcamd::check(kc_set_erase(kMers->h, w, complements ? 2 : 1, nullptr));
    kcamd::refresh(kMers);
}

/// Return the next k-mer of the set and update the index; kmer_t(-1) when exhausted (khash_utils.h:120-129).  The
/// reference scans its buckets upward from lastIndex; here the iteration order is the order of the first export of the
/// set (any fixed order serves: callers erase while they iterate and only need every live key exactly once).
template <typename kmer_t, typename kh_S_t>
kmer_t nextKMer(kh_S_t *kMers, kmer_t, size_t &lastIndex) {
    if (!kMers->h) return ~kmer_t(0);
    if (!kMers->have_order) {
        kcamd::refresh(kMers);
        const int W = kcamd::words_of<kmer_t>::value;
        std::vector<uint64_t> buf(std::max<uint64_t>(1, kMers->size * W));
        kcamd::check(kc_set_export(kMers->h, buf.data()));
        kMers->order.resize(kMers->size);
        for (uint64_t i = 0; i < kMers->size; ++i) kcamd::from_words(buf.data() + i * W, kMers->order[i]);
        kMers->have_order = true;
    }
    for (; lastIndex < kMers->order.size(); ++lastIndex) {
        uint64_t w[4] = {0, 0, 0, 0};
        kcamd::to_words(kMers->order[lastIndex], w);
        uint8_t present = 0;
        kcamd::check(kc_set_contains(kMers->h, w, 1, &present));
        if (present) return kMers->order[lastIndex];
    }
    return ~kmer_t(0);
}

/// Insert the k-mers of the vector (khash_utils.h:154-159).
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
void kMersFromVec(kh_S_t *kMers, kh_wrapper_t wrapper, std::vector<kmer_t> &kMerVec) {
    for (kmer_t kMer : kMerVec) { int ret; wrapper.kh_put_to_set(kMers, kMer, &ret); }
}

/// Construct a vector of the k-mer set in an arbitrary order (khash_utils.h:131-141).
template <typename kmer_t, typename kh_S_t>
std::vector<kmer_t> kMersToVec(kh_S_t *kMers, kmer_t) {
    std::vector<kmer_t> res;
    if (!kMers->h) return res;
    kcamd::refresh(kMers);
    const int W = kcamd::words_of<kmer_t>::value;
    std::vector<uint64_t> buf(std::max<uint64_t>(1, kMers->size * W));
    kcamd::check(kc_set_export(kMers->h, buf.data()));
    res.resize(kMers->size);
    for (uint64_t i = 0; i < kMers->size; ++i) kcamd::from_words(buf.data() + i * W, res[i]);
    return res;
}
