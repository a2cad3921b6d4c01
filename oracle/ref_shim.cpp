// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or executed from the product path.
//
// ref_shim.cpp: a thin driver (own code) around the UNMODIFIED reference headers, included where they lie
// under /root/reference/src (-I on the compiler command line, see oracle/Makefile).  Only the hot-path headers
// are pulled in (kmers.h, khash.h, khash_utils.h, parser.h, simplitigs.h, global.h, global_sparse.h,
// lower_bound.h); they compile as they are with g++ + zlib.  src/main.cpp is NOT buildable here without
// stand-ins (it needs the generated version.h and <glpk.h> through masks.h), so it is not built; the greedy
// branch of its driver kmercamel<>() (main.cpp:122-212) is re-driven below by calling the same reference
// functions in the same order.
//
// Outputs: oracle/_ref/libkcref.so (C entry points for ctypes) and oracle/_ref/kcref (CLI, -DKCREF_MAIN).
//
// Header-order note: ac/kmers_ac.h must be seen before kmers.h.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <sstream>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <algorithm>
#include <chrono>

#include "ac/kmers_ac.h"
#include "kmers.h"
#include "khash_utils.h"
#include "parser.h"
#include "simplitigs.h"
#include "global.h"
#include "global_sparse.h"
#include "lower_bound.h"
#include "conversions.h"

namespace {

constexpr int kSimplitigRatioThreshold = 5;  // main.cpp:94

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename kmer_t>
void kmer_to_words(kmer_t x, uint64_t w[4]);
template <> void kmer_to_words<uint64_t>(uint64_t x, uint64_t w[4]) { w[0] = x; w[1] = w[2] = w[3] = 0; }
template <> void kmer_to_words<__uint128_t>(__uint128_t x, uint64_t w[4]) {
    w[0] = (uint64_t)x; w[1] = (uint64_t)(x >> 64); w[2] = w[3] = 0;
}
template <> void kmer_to_words<uint256_t>(uint256_t x, uint64_t w[4]) {
    __uint128_t lo = x.lower(), hi = x.upper();
    w[0] = (uint64_t)lo; w[1] = (uint64_t)(lo >> 64); w[2] = (uint64_t)hi; w[3] = (uint64_t)(hi >> 64);
}
template <typename kmer_t>
kmer_t words_to_kmer(const uint64_t w[4]);
template <> uint64_t words_to_kmer<uint64_t>(const uint64_t w[4]) { return w[0]; }
template <> __uint128_t words_to_kmer<__uint128_t>(const uint64_t w[4]) { return ((__uint128_t)w[1] << 64) | w[0]; }
template <> uint256_t words_to_kmer<uint256_t>(const uint64_t w[4]) {
    __uint128_t lo = ((__uint128_t)w[1] << 64) | w[0], hi = ((__uint128_t)w[3] << 64) | w[2];
    return uint256_t(hi, lo);
}

std::string simplitig_to_string(simplitig_t &s) {
    std::string r(s.size() / 2, 'N');
    for (size_t i = 0; i < r.size(); ++i) r[i] = simplitig_at_index(s, i);
    return r;
}

struct Stats {
    uint64_t n_kmers = 0, n_simplitigs = 0, length = 0, sparse = 0;
    double t_read = 0, t_simplitigs = 0, t_path = 0, t_emit = 0;
};

// Greedy branch of kmercamel<>() (main.cpp:122-212), sequence line only (no FASTA header).
template <typename kmer_t, typename kh_wrapper_t>
int run_compute(kh_wrapper_t wrapper, kmer_t kt, std::string path, int k, bool complements, bool assume_simplitigs,
                std::ostream &of, std::ostream *maskf, Stats &st, std::vector<std::string> *simplitig_dump,
                overlapPath *path_dump, uint16_t min_frequency = 1) {
    auto *kMers = wrapper.kh_init_set();
    size_t kmer_count = 0;
    double t0 = now_s();
    if (!assume_simplitigs) {
        if (min_frequency == 1) ReadKMers(kMers, wrapper, kt, path, k, complements);
        else ReadKMersFiltered(kMers, wrapper, kt, path, k, complements, min_frequency);   // main.cpp:150-153
        if (!kh_size(kMers)) return 1;
        kmer_count = kh_size(kMers);
    }
    double t1 = now_s();
    std::vector<simplitig_t> simplitigs;
    if (!assume_simplitigs) {
        simplitigs = get_simplitigs(kMers, wrapper, kt, k, complements);
        wrapper.kh_destroy_set(kMers);
    } else {
        simplitigs = simplitigs_from_fasta(path);
    }
    double t2 = now_s();
    st.n_kmers = kmer_count;
    st.n_simplitigs = simplitigs.size();
    if (simplitig_dump) for (auto &s : simplitigs) simplitig_dump->push_back(simplitig_to_string(s));
    if (assume_simplitigs) {
        for (auto &s : simplitigs) st.n_kmers += kmers_in_simplitig(s, k);
    }
    double t3, t4;
    if (!assume_simplitigs && simplitigs.size() * kSimplitigRatioThreshold >= kmer_count) {
        st.sparse = 1;
        auto kMerVec = simplitigs_to_kmer_vec(kt, simplitigs, k, kmer_count);
        PartialPreSort(kMerVec, k);
        if (kMerVec.empty()) throw std::invalid_argument("input cannot be empty");
        auto hp = OverlapHamiltonianPathSparse(wrapper, kMerVec, k, complements);
        t3 = now_s();
        if (path_dump) *path_dump = hp;
        st.length = SuperstringFromPathSparse(wrapper, hp, kMerVec, of, maskf, k, complements);
        t4 = now_s();
    } else {
        if (simplitigs.empty()) throw std::invalid_argument("input cannot be empty");
        auto hp = OverlapHamiltonianPath(wrapper, kt, simplitigs, k, complements);
        t3 = now_s();
        if (path_dump) *path_dump = hp;
        st.length = SuperstringFromPath(wrapper, kt, hp, simplitigs, of, maskf, k, complements);
        t4 = now_s();
    }
    st.t_read = t1 - t0; st.t_simplitigs = t2 - t1; st.t_path = t3 - t2; st.t_emit = t4 - t3;
    return 0;
}

template <typename F>
auto dispatch_k(int k, F &&f) {
    if (k < 32) return f(kmer_dict64_t(), kmer64_t(0));
    if (k < 64) return f(kmer_dict128_t(), kmer128_t(0));
    return f(kmer_dict256_t(), kmer256_t(0));
}

char *dup_string(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}

std::vector<simplitig_t> strings_to_simplitigs(const char **strs, uint64_t n) {
    std::vector<simplitig_t> v;
    v.reserve(n);
    for (uint64_t i = 0; i < n; ++i) v.push_back(simplitig_from_string(std::string(strs[i])));
    return v;
}

}  // namespace

extern "C" {

void kcref_free(void *p) { free(p); }

// stats[8] = {n_kmers, n_simplitigs, length, sparse_branch, t_read_us, t_simplitigs_us, t_path_us, t_emit_us}
// Writes the sequence line(s) (no FASTA header, trailing '\n' as the reference prints them) to out_path / mask_path.
int kcref_compute_file(const char *path, int k, int complements, int assume_simplitigs, const char *out_path,
                       const char *mask_path, uint64_t *stats) {
    try {
        std::ofstream of(out_path);
        std::ofstream mf;
        std::ostream *maskf = nullptr;
        if (mask_path && mask_path[0]) { mf.open(mask_path); maskf = &mf; }
        Stats st;
        int rc = dispatch_k(k, [&](auto wrapper, auto kt) {
            return run_compute(wrapper, kt, std::string(path), k, complements != 0, assume_simplitigs != 0, of, maskf, st,
                               nullptr, nullptr);
        });
        of << std::endl;  // main.cpp:210
        if (stats) {
            stats[0] = st.n_kmers; stats[1] = st.n_simplitigs; stats[2] = st.length; stats[3] = st.sparse;
            stats[4] = (uint64_t)(st.t_read * 1e6); stats[5] = (uint64_t)(st.t_simplitigs * 1e6);
            stats[6] = (uint64_t)(st.t_path * 1e6); stats[7] = (uint64_t)(st.t_emit * 1e6);
        }
        return rc;
    } catch (std::exception &e) {
        std::cerr << "kcref: " << e.what() << std::endl;
        return -1;
    }
}

// compute with -z (ReadKMersFiltered, parser.h:122-141)
int kcref_compute_file_z(const char *path, int k, int complements, int min_frequency, const char *out_path, const char *mask_path,
                         uint64_t *stats) {
    try {
        std::ofstream of(out_path);
        std::ofstream mf;
        std::ostream *maskf = nullptr;
        if (mask_path && mask_path[0]) { mf.open(mask_path); maskf = &mf; }
        Stats st;
        int rc = dispatch_k(k, [&](auto wrapper, auto kt) {
            return run_compute(wrapper, kt, std::string(path), k, complements != 0, false, of, maskf, st, nullptr, nullptr, (uint16_t)min_frequency);
        });
        of << std::endl;
        if (stats) { stats[0] = st.n_kmers; stats[1] = st.n_simplitigs; stats[2] = st.length; stats[3] = st.sparse; }
        return rc;
    } catch (std::exception &e) {
        std::cerr << "kcref: " << e.what() << std::endl;
        return -1;
    }
}

// Number of k-mers ReadKMersFiltered keeps (tests/parser_unittest.h:64-91)
int64_t kcref_read_kmers_filtered_count(const char *path, int k, int complements, int min_frequency) {
    try {
        return dispatch_k(k, [&](auto wrapper, auto kt) -> int64_t {
            auto *set = wrapper.kh_init_set();
            std::string p(path);
            ReadKMersFiltered(set, wrapper, kt, p, k, complements != 0, (uint16_t)min_frequency);
            int64_t n = kh_size(set);
            wrapper.kh_destroy_set(set);
            return n;
        });
    } catch (std::exception &e) { return -1; }
}

// Sorted canonical k-mer set of one sequence through AddKMers (parser.h:22-49). Returns count, *out = malloc'd
// count*4 little-endian 64-bit words.
int64_t kcref_kmer_set(const char *seq, uint64_t len, int k, int complements, int case_sensitive, uint64_t **out) {
    return dispatch_k(k, [&](auto wrapper, auto kt) -> int64_t {
        using kmer_t = decltype(kt);
        auto *set = wrapper.kh_init_set();
        AddKMers(set, wrapper, kt, len, seq, k, complements != 0, case_sensitive != 0);
        auto vec = kMersToVec(set, kt);
        wrapper.kh_destroy_set(set);
        std::sort(vec.begin(), vec.end());
        uint64_t *w = (uint64_t *)malloc(std::max<size_t>(1, vec.size()) * 4 * sizeof(uint64_t));
        for (size_t i = 0; i < vec.size(); ++i) kmer_to_words<kmer_t>(vec[i], w + 4 * i);
        *out = w;
        return (int64_t)vec.size();
    });
}

// Number of k-mers ReadKMers collects from a file (tests/parser_unittest.h:38-63).
int64_t kcref_read_kmers_count(const char *path, int k, int complements, int case_sensitive) {
    try {
        return dispatch_k(k, [&](auto wrapper, auto kt) -> int64_t {
            auto *set = wrapper.kh_init_set();
            std::string p(path);
            ReadKMers(set, wrapper, kt, p, k, complements != 0, case_sensitive != 0);
            int64_t n = kh_size(set);
            wrapper.kh_destroy_set(set);
            return n;
        });
    } catch (std::exception &e) { return -1; }
}

// get_simplitigs (simplitigs.h:191-205) on the set built from `seq` by AddKMers. Returns '\n'-joined simplitigs in
// the reference's own (hash-layout) order.
char *kcref_simplitigs_of_sequence(const char *seq, uint64_t len, int k, int complements) {
    std::string res = dispatch_k(k, [&](auto wrapper, auto kt) -> std::string {
        auto *set = wrapper.kh_init_set();
        AddKMers(set, wrapper, kt, len, seq, k, complements != 0, false);
        auto s = get_simplitigs(set, wrapper, kt, k, complements != 0);
        wrapper.kh_destroy_set(set);
        std::string r;
        for (auto &x : s) { r += simplitig_to_string(x); r += '\n'; }
        return r;
    });
    return dup_string(res);
}

// get_simplitigs on a set filled by kh_put in the given order (k-mers as 4-word little-endian records).
char *kcref_simplitigs_of_kmers(const uint64_t *words, uint64_t n, int k, int complements) {
    std::string res = dispatch_k(k, [&](auto wrapper, auto kt) -> std::string {
        using kmer_t = decltype(kt);
        auto *set = wrapper.kh_init_set();
        int ret;
        for (uint64_t i = 0; i < n; ++i) wrapper.kh_put_to_set(set, words_to_kmer<kmer_t>(words + 4 * i), &ret);
        auto s = get_simplitigs(set, wrapper, kt, k, complements != 0);
        wrapper.kh_destroy_set(set);
        std::string r;
        for (auto &x : s) { r += simplitig_to_string(x); r += '\n'; }
        return r;
    });
    return dup_string(res);
}

// OverlapHamiltonianPath (global.h:43-133) on simplitigs given as ACGT strings (simplitig_from_string).
int kcref_overlap_path(const char **simplitigs, uint64_t n, int k, int complements, int lower_bound, uint64_t *edge_from,
                       uint8_t *overlaps) {
    return dispatch_k(k, [&](auto wrapper, auto kt) -> int {
        auto v = strings_to_simplitigs(simplitigs, n);
        auto hp = OverlapHamiltonianPath(wrapper, kt, v, k, complements != 0, lower_bound != 0);
        for (size_t i = 0; i < hp.first.size(); ++i) { edge_from[i] = hp.first[i]; overlaps[i] = hp.second[i]; }
        return 0;
    });
}

// SuperstringFromPath (global.h:149-210). *out / *mask_out malloc'd (mask_out may be NULL). Returns length.
int64_t kcref_superstring_from_path(const char **simplitigs, uint64_t n, const uint64_t *edge_from, const uint8_t *overlaps,
                                    int k, int complements, char **out, char **mask_out) {
    return dispatch_k(k, [&](auto wrapper, auto kt) -> int64_t {
        auto v = strings_to_simplitigs(simplitigs, n);
        size_t V = n * (1 + (complements != 0));
        overlapPath hp;
        hp.first.assign(edge_from, edge_from + V);
        hp.second.assign(overlaps, overlaps + V);
        std::stringstream of, mf;
        size_t L = SuperstringFromPath(wrapper, kt, hp, v, of, mask_out ? &mf : nullptr, k, complements != 0);
        *out = dup_string(of.str());
        if (mask_out) *mask_out = dup_string(mf.str());
        return (int64_t)L;
    });
}

// Global (global.h:217-226) on ACGT strings; returns 0, -2 for "input cannot be empty".
int kcref_global(const char **simplitigs, uint64_t n, int k, int complements, char **out, char **mask_out) {
    try {
        return dispatch_k(k, [&](auto wrapper, auto kt) -> int {
            auto v = strings_to_simplitigs(simplitigs, n);
            std::stringstream of, mf;
            Global(wrapper, kt, v, of, mask_out ? &mf : nullptr, k, complements != 0);
            *out = dup_string(of.str());
            if (mask_out) *mask_out = dup_string(mf.str());
            return 0;
        });
    } catch (std::invalid_argument &e) { return -2; }
}

// GlobalSparse (global_sparse.h:211-219) on k-mers (4-word records) in the given order (after PartialPreSort if asked).
int kcref_global_sparse(const uint64_t *words, uint64_t n, int k, int complements, int presort, char **out, char **mask_out) {
    try {
        return dispatch_k(k, [&](auto wrapper, auto kt) -> int {
            using kmer_t = decltype(kt);
            std::vector<kmer_t> v(n);
            for (uint64_t i = 0; i < n; ++i) v[i] = words_to_kmer<kmer_t>(words + 4 * i);
            if (presort) PartialPreSort(v, k);
            std::stringstream of, mf;
            GlobalSparse(wrapper, v, of, mask_out ? &mf : nullptr, k, complements != 0);
            *out = dup_string(of.str());
            if (mask_out) *mask_out = dup_string(mf.str());
            return 0;
        });
    } catch (std::invalid_argument &e) { return -2; }
}

// LowerBoundLength (lower_bound.h:7-21).
int64_t kcref_lower_bound(const char **simplitigs, uint64_t n, int k, int complements) {
    return dispatch_k(k, [&](auto wrapper, auto kt) -> int64_t {
        auto v = strings_to_simplitigs(simplitigs, n);
        return (int64_t)LowerBoundLength(wrapper, kt, v, k, complements != 0);
    });
}

// k-mer primitives (kmers.h:36-44, 93-127) on 4-word little-endian records.
void kcref_reverse_complement(const uint64_t *in, int k, uint64_t *out) {
    dispatch_k(k, [&](auto, auto kt) -> int {
        using kmer_t = decltype(kt);
        kmer_to_words<kmer_t>(ReverseComplement(words_to_kmer<kmer_t>(in), k), out);
        return 0;
    });
}
void kcref_bit_prefix(const uint64_t *in, int k, int d, uint64_t *out) {
    dispatch_k(k, [&](auto, auto kt) -> int {
        using kmer_t = decltype(kt);
        kmer_to_words<kmer_t>(BitPrefix(words_to_kmer<kmer_t>(in), k, d), out);
        return 0;
    });
}
void kcref_bit_suffix(const uint64_t *in, int k, int d, uint64_t *out) {
    dispatch_k(k, [&](auto, auto kt) -> int {
        using kmer_t = decltype(kt);
        kmer_to_words<kmer_t>(BitSuffix(words_to_kmer<kmer_t>(in), d), out);
        return 0;
    });
}
// Representation converters (conversions.h:16-91); results are malloc'ed C strings released with kcref_free.
int kcref_split_ms(const char *path, char **superstring, char **mask) {
    try {
        std::ostringstream s, m;
        split_ms(s, m, std::string(path));
        *superstring = dup_string(s.str());
        *mask = dup_string(m.str());
        return 0;
    } catch (std::exception &) { return -1; }
}
int kcref_join_ms(const char *superstring_text, const char *mask_text, char **out) {
    std::istringstream s{std::string(superstring_text)}, m{std::string(mask_text)};
    std::ostringstream o;
    join_ms(s, m, o);
    *out = dup_string(o.str());
    return 0;
}
int kcref_ms_to_spss(const char *path, int k, char **out) {
    try {
        std::ostringstream o;
        ms_to_spss(std::string(path), o, k);
        *out = dup_string(o.str());
        return 0;
    } catch (std::exception &) { return -1; }
}
int kcref_spss_to_ms(const char *path, int k, char **out) {
    try {
        std::ostringstream o;
        spss_to_ms(std::string(path), o, k);
        *out = dup_string(o.str());
        return 0;
    } catch (std::exception &) { return -1; }
}

char kcref_masked(char c, int m) { return Masked(c, m != 0); }
int kcref_nucleotide_to_int(int c) { return nucleotideToInt[(uint8_t)c]; }

// Hash functions of the set (khash.h:390,417-426; khash_utils.h:10-13).
uint64_t kcref_hash(const uint64_t *in, int k) {
    if (k < 32) return kh_int64_hash_func(in[0]);
    if (k < 64) { __uint128_t x = words_to_kmer<__uint128_t>(in); return kh_int128_hash_func(x); }
    uint256_t x = words_to_kmer<uint256_t>(in);
    return kh_int256_hash_func(x);
}

}  // extern "C"

#ifdef KCREF_MAIN
// CLI: kcref compute -k K [-u] [-S] [-o out] [-M maskout] input   (sequence lines only + timing on stderr)
int main(int argc, char **argv) {
    if (argc < 3 || std::string(argv[1]) != "compute") {
        std::cerr << "usage: kcref compute -k K [-u] [-S] [-o out] [-M maskout] <fasta>" << std::endl;
        return 1;
    }
    int k = 0; bool complements = true, assume = false;
    std::string out = "/dev/stdout", mask, path = argv[argc - 1];
    for (int i = 2; i < argc - 1; ++i) {
        std::string a = argv[i];
        if (a == "-k") k = std::stoi(argv[++i]);
        else if (a == "-u") complements = false;
        else if (a == "-S") assume = true;
        else if (a == "-o") out = argv[++i];
        else if (a == "-M") mask = argv[++i];
    }
    uint64_t st[8];
    double t0 = now_s();
    int rc = kcref_compute_file(path.c_str(), k, complements, assume, out.c_str(), mask.c_str(), st);
    double t1 = now_s();
    std::cerr << "{\"n_kmers\": " << st[0] << ", \"n_simplitigs\": " << st[1] << ", \"length\": " << st[2]
              << ", \"sparse\": " << st[3] << ", \"t_read_s\": " << st[4] / 1e6 << ", \"t_simplitigs_s\": " << st[5] / 1e6
              << ", \"t_path_s\": " << st[6] / 1e6 << ", \"t_emit_s\": " << st[7] / 1e6 << ", \"t_total_s\": " << (t1 - t0)
              << "}" << std::endl;
    return rc;
}
#endif
