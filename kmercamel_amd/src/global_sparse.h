// Drop-in for the reference's src/global_sparse.h: the overlap graph over single k-mers that kmercamel<>() falls back to when
// there are too many simplitigs (main.cpp:175-181).  Same entry points and signatures (global_sparse.h:14, 43, 138, 211);
// a k-mer is a simplitig of length k, so the calls go through the same device path as global.h.
#pragma once
#include <vector>
#include <iostream>
#include "global.h"

constexpr int SORT_FIRST_BITS_DEFAULT = 8;

/// Stable bucketing of the k-mers by their leading min(2k, 8) bits (global_sparse.h:14-35).
template <typename kmer_t>
void PartialPreSort(std::vector<kmer_t> &vals, int k) {
    const int bits = 2 * k < SORT_FIRST_BITS_DEFAULT ? 2 * k : SORT_FIRST_BITS_DEFAULT, shift = 2 * k - bits;
    std::vector<std::vector<kmer_t>> bucket((size_t)1 << bits);
    for (auto &x : vals) bucket[(size_t)(kcamd::low64(x >> shift) & (((uint64_t)1 << bits) - 1))].push_back(x);
    size_t i = 0;
    for (auto &b : bucket) for (auto &x : b) vals[i++] = x;
}

namespace kcamd {
template <typename kmer_t> inline std::vector<simplitig_t> kmers_as_simplitigs(const std::vector<kmer_t> &kMers, int k) {
    std::vector<simplitig_t> v;
    v.reserve(kMers.size());
    for (auto &x : kMers) v.push_back(simplitig_from_string(NumberToKMer(x, k)));
    return v;
}
}  // namespace kcamd

template <typename kmer_t, typename kh_wrapper_t>
overlapPath OverlapHamiltonianPathSparse(kh_wrapper_t wrapper, std::vector<kmer_t> &kMers, int k, bool complements, bool lower_bound = false) {
    auto v = kcamd::kmers_as_simplitigs(kMers, k);
    return OverlapHamiltonianPath(wrapper, kmer_t(0), v, k, complements, lower_bound);
}

template <typename kmer_t, typename kh_wrapper_t>
size_t SuperstringFromPathSparse(kh_wrapper_t wrapper, const overlapPath &hamiltonianPath, const std::vector<kmer_t> &kMers, std::ostream &of,
                                 std::ostream *maskf, const int k, const bool complements) {
    auto v = kcamd::kmers_as_simplitigs(kMers, k);
    return SuperstringFromPath(wrapper, kmer_t(0), hamiltonianPath, v, of, maskf, k, complements);
}

template <typename kmer_t, typename kh_wrapper_t>
void GlobalSparse(kh_wrapper_t wrapper, std::vector<kmer_t> &kMers, std::ostream &of, std::ostream *maskf, int k, bool complements) {
    if (kMers.empty()) throw std::invalid_argument("input cannot be empty");
    auto v = kcamd::kmers_as_simplitigs(kMers, k);
    Global(wrapper, kmer_t(0), v, of, maskf, k, complements);
}
