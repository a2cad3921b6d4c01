// Drop-in for the reference's src/global.h entry points with identical signatures (global.h:35, 43-45, 149-150, 217-218):
// overlapPath, OverlapHamiltonianPath, SuperstringFromPath, Global.  Each is a thin shim over the C ABI.
#pragma once
#include <vector>
#include <iostream>
#include <utility>
#include "kmers.h"
#include "khash_utils.h"
#include "simplitigs.h"

typedef std::pair<std::vector<size_t>, std::vector<unsigned char>> overlapPath;

template <typename kmer_t, typename kh_wrapper_t>
overlapPath OverlapHamiltonianPath(kh_wrapper_t, kmer_t, std::vector<simplitig_t> &simplitigs, int k, bool complements, bool lower_bound = false) {
    size_t V = simplitigs.size() * (1 + complements);
    overlapPath res{std::vector<size_t>(V, (size_t)-1), std::vector<unsigned char>(V, (unsigned char)-1)};
    if (simplitigs.empty()) return res;
    kc_simplitigs *h = kcamd::upload_simplitigs(simplitigs, k);
    std::vector<uint64_t> ef(V);
    int rc = kc_overlap_path(h, complements, lower_bound, ef.data(), res.second.data());
    kc_simplitigs_destroy(h);
    kcamd::check(rc);
    for (size_t i = 0; i < V; ++i) res.first[i] = (size_t)ef[i];
    return res;
}

template <typename kmer_t, typename kh_wrapper_t>
size_t SuperstringFromPath(kh_wrapper_t, kmer_t, const overlapPath &hamiltonianPath, const std::vector<simplitig_t> &simplitigs, std::ostream &of,
                           std::ostream *maskf, const int k, const bool complements) {
    kc_simplitigs *h = kcamd::upload_simplitigs(simplitigs, k);
    std::vector<uint64_t> ef(hamiltonianPath.first.begin(), hamiltonianPath.first.end());
    kc_set *set = nullptr;
    char *out = nullptr, *mask = nullptr;
    uint64_t L = 0;
    int rc = KC_OK;
    if (maskf != nullptr) {   // fill_kmers (simplitigs.h:41-51): the set of all k-mers of the simplitigs
        rc = kc_set_create(kcamd::context(), k, complements, 0, &set);
        for (size_t i = 0; rc == KC_OK && i < simplitigs.size(); ++i) {
            std::string s = simplitig_to_string(simplitigs[i]);
            rc = kc_set_add_sequence(set, s.data(), s.size(), 0);
        }
    }
    if (rc == KC_OK) rc = kc_superstring_from_path(h, ef.data(), hamiltonianPath.second.data(), complements, set, &out, maskf ? &mask : nullptr, &L);
    if (set) kc_set_destroy(set);
    kc_simplitigs_destroy(h);
    kcamd::check(rc);
    of << out;
    if (maskf != nullptr) (*maskf) << mask << std::endl;   // global.h:206-208
    kc_free(out);
    kc_free(mask);
    return (size_t)L;
}

/// Global greedy (global.h:217-226): path + superstring stay on the device in between.
template <typename kmer_t, typename kh_wrapper_t>
void Global(kh_wrapper_t, kmer_t, std::vector<simplitig_t> &simplitigs, std::ostream &of, std::ostream *maskf, int k, bool complements) {
    if (simplitigs.empty()) throw std::invalid_argument("input cannot be empty");
    kc_simplitigs *h = kcamd::upload_simplitigs(simplitigs, k);
    char *out = nullptr, *mask = nullptr;
    uint64_t L = 0;
    int rc = kc_global(h, complements, nullptr, maskf != nullptr, &out, maskf ? &mask : nullptr, &L);
    kc_simplitigs_destroy(h);
    kcamd::check(rc);
    WriteLog("Finished 2. part: Hamiltonian path.");
    of << out;
    if (maskf != nullptr) (*maskf) << mask << std::endl;
    kc_free(out);
    kc_free(mask);
    WriteLog("Finished 3. part: masked superstring (l=" + std::to_string(L) + ").");
}
