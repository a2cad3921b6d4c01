// Drop-in for the reference's src/lower_bound.h (lower_bound.h:7-36): length of the cycle cover of the overlap graph, a
// lower bound of the superstring length.  The cycle cover is phase 2 with lower_bound = true on the device (kc_lower_bound).
#pragma once
#include "global.h"
#include "global_sparse.h"

template <typename kmer_t, typename kh_wrapper_t>
size_t LowerBoundLength(kh_wrapper_t, kmer_t, std::vector<simplitig_t> simplitigs, int k, bool complements) {
    if (simplitigs.empty()) return 0;
    kc_simplitigs *h = kcamd::upload_simplitigs(simplitigs, k);
    uint64_t res = 0;
    int rc = kc_lower_bound(h, complements, &res);
    kc_simplitigs_destroy(h);
    kcamd::check(rc);
    WriteLog("Finished 2. part: Hamiltonian path.");
    WriteLog("Finished 3. part: lower bound = " + std::to_string(res) + ".");
    return (size_t)res;
}

template <typename kmer_t, typename kh_wrapper_t>
size_t LowerBoundLengthSparse(kh_wrapper_t wrapper, std::vector<kmer_t> &kMers, int k, bool complements) {
    return LowerBoundLength(wrapper, kmer_t(0), kcamd::kmers_as_simplitigs(kMers, k), k, complements);
}
