// Drop-in for the reference's src/simplitigs.h entry points: simplitig_t, get_simplitigs, next_simplitig,
// simplitigs_to_kmer_vec, simplitigs_from_fasta and the small accessors (simplitigs.h:11-103, 142-205).  std::vector<bool> simplitigs are converted to / from the packed
// device collection at this boundary only; the CLI's fused path never materialises them.
#pragma once
#include <vector>
#include <string>
#include <cassert>
#include "kmers.h"
#include "khash_utils.h"
#include "parser.h"

typedef std::vector<bool> simplitig_t;   // 2 bits per nucleotide, high bit first (simplitigs.h:11,23-25)

inline simplitig_t simplitig_reverse_complement(simplitig_t forward) {   // simplitigs.h:14-21
    simplitig_t reverse(forward.size());
    size_t n = forward.size() / 2;
    for (size_t i = 0; i < n; ++i) {
        int code = 3 - (forward[2 * i] * 2 + forward[2 * i + 1]);
        reverse[2 * (n - 1 - i)] = code & 2;
        reverse[2 * (n - 1 - i) + 1] = code & 1;
    }
    return reverse;
}
inline char simplitig_at_index(simplitig_t &simplitig, size_t index) { return letters[simplitig[2 * index] * 2 + simplitig[2 * index + 1]]; }
inline size_t kmers_in_simplitig(simplitig_t &simplitig, int k) { return simplitig.size() / 2 - k + 1; }
template <typename kmer_t> kmer_t kmer_at_simplitig_index(kmer_t, simplitig_t &simplitig, int k, size_t index) {
    kmer_t ret = 0;
    for (size_t i = 2 * index; i < 2 * (index + k); ++i) { ret = ret << 1; ret = ret | kmer_t((uint64_t)simplitig[i]); }
    return ret;
}
template <typename kmer_t> kmer_t simplitig_first(kmer_t t, simplitig_t &s, int k) { return kmer_at_simplitig_index(t, s, k, 0); }
template <typename kmer_t> kmer_t simplitig_last(kmer_t t, simplitig_t &s, int k) { return kmer_at_simplitig_index(t, s, k, kmers_in_simplitig(s, k) - 1); }

inline simplitig_t simplitig_from_string(std::string sequence) {   // simplitigs.h:78-87
    simplitig_t ret(2 * sequence.size());
    for (size_t i = 0; i < sequence.size(); ++i) {
        uint8_t x = nucleotideToInt[(uint8_t)sequence[i]];
        assert(x <= 3);
        ret[2 * i] = x & 2;
        ret[2 * i + 1] = x & 1;
    }
    return ret;
}
inline std::string simplitig_to_string(const simplitig_t &s) {
    std::string r(s.size() / 2, 'N');
    for (size_t i = 0; i < r.size(); ++i) r[i] = letters[s[2 * i] * 2 + s[2 * i + 1]];
    return r;
}

namespace kcamd {
// std::vector<simplitig_t> -> device collection
inline kc_simplitigs *upload_simplitigs(const std::vector<simplitig_t> &v, int k) {
    std::string cat;
    std::vector<uint64_t> off(v.size() + 1);
    for (size_t i = 0; i < v.size(); ++i) { off[i] = cat.size(); cat += simplitig_to_string(v[i]); }
    off[v.size()] = cat.size();
    kc_simplitigs *h = nullptr;
    check(kc_simplitigs_from_strings(context(), k, cat.data(), off.data(), v.size(), &h));
    return h;
}
inline std::vector<simplitig_t> download_simplitigs(kc_simplitigs *h) {
    uint64_t n = 0, total = 0;
    check(kc_simplitigs_count(h, &n, &total));
    std::string cat(total, 'N');
    std::vector<uint64_t> off(n + 1);
    check(kc_simplitigs_export(h, cat.data(), off.data()));
    std::vector<simplitig_t> out;
    out.reserve(n);
    for (uint64_t i = 0; i < n; ++i) out.push_back(simplitig_from_string(cat.substr(off[i], off[i + 1] - off[i])));
    return out;
}
}  // namespace kcamd

inline std::vector<simplitig_t> simplitigs_from_fasta(std::string &path) {   // simplitigs.h:89-103
    std::vector<simplitig_t> simplitigs;
    for (auto &r : kcamd::read_records(path)) simplitigs.push_back(simplitig_from_string(r.seq));
    return simplitigs;
}

/// All k-mers of the simplitigs, in order (simplitigs.h:53-65).
template <typename kmer_t>
std::vector<kmer_t> simplitigs_to_kmer_vec(kmer_t kmer_type, std::vector<simplitig_t> simplitigs, int k, size_t length_estimate) {
    std::vector<kmer_t> result;
    result.reserve(length_estimate);
    for (auto &s : simplitigs)
        for (size_t i = 0, n = kmers_in_simplitig(s, k); i < n; ++i) result.push_back(kmer_at_simplitig_index(kmer_type, s, k, i));
    return result;
}

/// One seed-and-extend simplitig in the reference's own sequential manner (simplitigs.h:105-189): the seed and every
/// k-mer the walk takes are removed from the set; the right end tries T, G, C, A, the left end A, C, G, T.  This is the
/// element-wise compatibility entry point -- every step is a handful of single-key device lookups; whole sets go through
/// get_simplitigs, which computes all simplitigs in a few kernels.
namespace kcamd {
// one step at the end whose outward-reading k-mer is `fw` (and `bw` its reverse complement; uni-directional walks keep only
// the strand they move on and mark the other with all ones): returns the extension code 0..3 or 0xFF
template <bool complements, typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
inline uint8_t step_outwards(kmer_t &fw, kmer_t &bw, kh_S_t *kMers, kh_wrapper_t &wrapper, int k) {
    const bool on_forward = complements || bw == ~kmer_t(0);
    const kmer_t mask = (kmer_t(1) << (2 * k)) - kmer_t(1);
    for (int ext = 0; ext < 4; ++ext) {
        const kmer_t nf = ((fw << 2) & mask) | kmer_t((uint64_t)(3 ^ ext));
        const kmer_t nb = (bw >> 2) | (kmer_t((uint64_t)ext) << (2 * (k - 1)));
        const kmer_t probe = complements ? (nf <= nb ? nf : nb) : (on_forward ? nf : nb);
        const khint_t it = wrapper.kh_get_from_set(kMers, probe);
        if (it == kh_end(kMers)) continue;
        if (complements) { fw = nf; bw = nb; }
        else if (on_forward) fw = nf;
        else bw = nb;
        wrapper.kh_del_from_set(kMers, it);
        return (uint8_t)ext;
    }
    return 0xFF;
}
}  // namespace kcamd
template <bool complements, typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
simplitig_t next_simplitig(kh_S_t *kMers, kh_wrapper_t wrapper, kmer_t begin, int k) {
    kmer_t last = begin, first = begin;
    kmer_t last_rc = complements ? ReverseComplement(begin, k) : ~kmer_t(0), first_rc = last_rc;
    simplitig_t right(2 * k), left;
    for (int i = 0; i < 2 * k; ++i) right[2 * k - 1 - i] = (kcamd::low64(begin >> i) & 1) != 0;
    eraseKMer(kMers, wrapper, begin, k, complements);
    for (uint8_t e; (e = kcamd::step_outwards<complements>(last, last_rc, kMers, wrapper, k)) != 0xFF;) {
        right.push_back(!(e & 2));   // the appended nucleotide is 3 ^ e
        right.push_back(!(e & 1));
    }
    for (uint8_t e; (e = kcamd::step_outwards<complements>(first_rc, first, kMers, wrapper, k)) != 0xFF;) {
        left.push_back(e & 1);       // the prepended nucleotide is e; pushed low bit first, the whole prefix is reversed below
        left.push_back(e & 2);
    }
    simplitig_t res(left.rbegin(), left.rend());
    res.insert(res.end(), right.begin(), right.end());
    return res;
}

/// Greedy simplitigs of the k-mer set (simplitigs.h:191-205).  Unlike the reference the set is left intact.
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
std::vector<simplitig_t> get_simplitigs(kh_S_t *kMers, kh_wrapper_t, kmer_t, int k, bool complements) {
    std::vector<simplitig_t> res;
    if (!kMers->h || kMers->size == 0) return res;
    if (kMers->k != k || kMers->complements != complements)
        throw std::invalid_argument("get_simplitigs: the set was filled with a different k / strand mode");
    kc_simplitigs *h = nullptr;
    kcamd::check(kc_simplitigs_build(kMers->h, &h));
    res = kcamd::download_simplitigs(h);
    kc_simplitigs_destroy(h);
    return res;
}
