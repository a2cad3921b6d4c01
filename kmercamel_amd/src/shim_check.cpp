// Exercises the drop-in templates exactly the way the reference's driver and unit tests call them
// (main.cpp:146-183; tests/global_unittest.h, tests/simplitigs_unittest.h, tests/parser_unittest.h) for all three key
// widths.  Run on a GPU box: prints PASS/FAIL lines, exit code = number of failures.  (tests/test_gpu_cli.py runs it.)
#include <iostream>
#include <sstream>
#include <algorithm>
#include "global.h"
#include "global_sparse.h"
#include "lower_bound.h"
#include "masks.h"
#include "parser.h"

static int failures = 0;
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { ++failures; std::cout << "FAIL " << __FILE__ << ":" << __LINE__ << " " #a " != " #b << std::endl; } } while (0)

template <typename kmer_t> kmer_t KM(const std::string &s) {   // KMerToNumber (tests use it to spell keys)
    kmer_t r = kmer_t(0);
    for (char c : s) r = (r << 2) | kmer_t((uint64_t)nucleotideToInt[(uint8_t)c]);
    return r;
}

template <typename kmer_t, typename wrapper_t, typename set_t> void run(const char *name, const std::string &testfa) {
    wrapper_t wrapper;
    typedef unsigned char byte;
    {   // next_simplitig, eraseKMer, nextKMer (tests/simplitigs_unittest.h:142-181; khash_utils.h:105-129)
        struct T { std::vector<kmer_t> kmers; int k; bool c; std::string want; std::vector<kmer_t> left; };
        std::vector<T> tests = {{{kmer_t(0b00010000), kmer_t(0b00000100)}, 4, false, "AACAA", {}},
                                {{kmer_t(0b00010000), kmer_t(0b00000011), kmer_t(0b00000100)}, 4, true, "AACAA", {kmer_t(0b00000011)}}};
        for (auto &t : tests) {
            auto kmers = wrapper.kh_init_set();
            int ret;
            for (auto &x : t.kmers) wrapper.kh_put_to_set(kmers, x, &ret);
            simplitig_t got = t.c ? next_simplitig<true>(kmers, wrapper, t.kmers.front(), t.k) : next_simplitig<false>(kmers, wrapper, t.kmers.front(), t.k);
            auto rest = kMersToVec(kmers, kmer_t(0));
            std::sort(rest.begin(), rest.end());
            EXPECT_EQ(t.left, rest);
            simplitig_t want = simplitig_from_string(t.want);
            bool ok = want == got || (t.c && simplitig_reverse_complement(want) == got);
            EXPECT_EQ(ok, true);
            wrapper.kh_destroy_set(kmers);
        }
        auto kmers = wrapper.kh_init_set();
        int ret;
        std::vector<kmer_t> in = {KM<kmer_t>("GCT"), KM<kmer_t>("TAA"), KM<kmer_t>("AAA"), KM<kmer_t>("CCG")};
        for (auto &x : in) wrapper.kh_put_to_set(kmers, x, &ret);
        EXPECT_EQ(containsKMer(kmers, wrapper, KM<kmer_t>("AGC"), 3, true), true);     // RC of GCT
        eraseKMer(kmers, wrapper, KM<kmer_t>("AGC"), 3, true);                         // removes the k-mer and its reverse complement
        EXPECT_EQ((size_t)kh_size(kmers), (size_t)3);
        std::vector<kmer_t> seen;
        size_t idx = 0;
        for (kmer_t x; !((x = nextKMer(kmers, kmer_t(0), idx)) == ~kmer_t(0)); ++idx) { seen.push_back(x); if (seen.size() == 1) eraseKMer(kmers, wrapper, KM<kmer_t>("CCG"), 3, false); }
        std::sort(seen.begin(), seen.end());
        // CCG was erased during the iteration: it is reported only if it came first
        bool iter_ok = (seen.size() == 2 && !(seen[0] == KM<kmer_t>("CCG")) && !(seen[1] == KM<kmer_t>("CCG"))) || (seen.size() == 3);
        EXPECT_EQ(iter_ok, true);
        wrapper.kh_destroy_set(kmers);
    }
    {   // sparse branch (tests/global_sparse_unittest.h:11-135)
        std::vector<kmer_t> v = {KM<kmer_t>("GTA"), KM<kmer_t>("TAC"), KM<kmer_t>("GGC")}, w = {KM<kmer_t>("GGC"), KM<kmer_t>("GTA"), KM<kmer_t>("TAC")};
        PartialPreSort(v, 3);
        EXPECT_EQ(v, w);
        std::vector<kmer_t> p13 = {KM<kmer_t>("AAAAAAAAAAAAT"), KM<kmer_t>("AAAAAAAAAAAAA")}, q13 = p13;
        PartialPreSort(p13, 13);
        EXPECT_EQ(p13, q13);
        struct P { overlapPath path; std::vector<kmer_t> kmers; int k; std::string want; bool c; };
        std::vector<P> ps = {{{{2, 0, (size_t)-1}, {1, 2, (byte)-1}}, {KM<kmer_t>("ACG"), KM<kmer_t>("TAC"), KM<kmer_t>("GGC")}, 3, "TAcGgc", false},
                             {{{4, 3, 1, (size_t)-1, 5, (size_t)-1}, {1, 2, 1, (byte)-1, 2, (byte)-1}}, {KM<kmer_t>("GCC"), KM<kmer_t>("ACG"), KM<kmer_t>("TAC")}, 3, "GcCGta", true}};
        for (auto &t : ps) {
            std::stringstream of;
            SuperstringFromPathSparse(wrapper, t.path, t.kmers, of, nullptr, t.k, t.c);
            EXPECT_EQ(t.want, of.str());
        }
        struct O { std::vector<kmer_t> kmers; overlapPath want; int k; bool c; };
        std::vector<O> os = {{{KM<kmer_t>("AT")}, {{(size_t)-1, (size_t)-1}, {(byte)-1, (byte)-1}}, 2, true},
                             {{KM<kmer_t>("ACG"), KM<kmer_t>("TAC"), KM<kmer_t>("GGC")}, {{2, 0, (size_t)-1}, {1, 2, (byte)-1}}, 3, false},
                             {{KM<kmer_t>("ACAA"), KM<kmer_t>("ATTT"), KM<kmer_t>("AACA")}, {{4, 3, 0, 5, (size_t)-1, (size_t)-1}, {2, 2, 3, 3, (byte)-1, (byte)-1}}, 4, true}};
        for (auto &t : os) {
            overlapPath got = OverlapHamiltonianPathSparse(wrapper, t.kmers, t.k, t.c);
            EXPECT_EQ(t.want.first, got.first);
            EXPECT_EQ(t.want.second, got.second);
        }
        struct G { std::string want; int k; std::vector<kmer_t> in; bool c; };
        std::vector<G> gs = {{"TACgt", 3, {KM<kmer_t>("CGT"), KM<kmer_t>("TAC"), KM<kmer_t>("ACG")}, false},
                             {"ACgTtt", 3, {KM<kmer_t>("CGT"), KM<kmer_t>("TTT"), KM<kmer_t>("ACG")}, false},
                             {"TActt", 4, {KM<kmer_t>("TACT"), KM<kmer_t>("ACTT")}, false},
                             {"TActTaaGgac", 4, {KM<kmer_t>("TACT"), KM<kmer_t>("ACTT"), KM<kmer_t>("GGAC"), KM<kmer_t>("TAAG")}, false},
                             {"TTtcttttttttttttttttttttttttttga", 31, {KM<kmer_t>("TTTCTTTTTTTTTTTTTTTTTTTTTTTTTTG"), KM<kmer_t>("TTCTTTTTTTTTTTTTTTTTTTTTTTTTTGA")}, false},
                             {"AtTTgtt", 4, {KM<kmer_t>("ACAA"), KM<kmer_t>("ATTT"), KM<kmer_t>("AACA")}, true}};
        for (auto &t : gs) {
            std::stringstream of;
            GlobalSparse(wrapper, t.in, of, nullptr, t.k, t.c);
            EXPECT_EQ(t.want, of.str());
        }
    }
    {   // LowerBoundLength (tests/lower_bound_unittest.h:11-52)
        struct T { std::vector<std::string> in; int k; bool c; size_t want; };
        std::vector<T> tests = {{{"ACG", "CGT", "TAA"}, 3, false, 5}, {{"ACGT", "TAA"}, 3, false, 5}, {{"AC", "GG"}, 2, true, 3}, {{"AAACCC", "CCCAAA", "TGGGGT"}, 6, false, 11}};
        for (auto &t : tests) {
            std::vector<simplitig_t> v;
            for (auto &s : t.in) v.push_back(simplitig_from_string(s));
            EXPECT_EQ(t.want, LowerBoundLength(wrapper, kmer_t(0), v, t.k, t.c));
        }
        std::vector<simplitig_t> v = {simplitig_from_string("ACG"), simplitig_from_string("CGT"), simplitig_from_string("TAA")};
        auto vec = simplitigs_to_kmer_vec(kmer_t(0), v, 3, 16);
        EXPECT_EQ(LowerBoundLengthSparse(wrapper, vec, 3, false), LowerBoundLength(wrapper, kmer_t(0), v, 3, false));
    }
    {   // Optimize max-one / min-one on tests/testdata/masktest.fa (tests/masks_unittest.h:19-65: the cases whose k-mer set
        // is the file's own represented set, which is what Optimize builds) + ReadKMersFiltered (tests/parser_unittest.h:64-91)
        std::string mpath = testfa.substr(0, testfa.find_last_of('/') + 1) + "masktest.fa";
        struct T { bool c; std::string alg, want; };   // masktest.fa = ACGTAACGT, every position switched on
        for (auto &t : std::vector<T>{{false, "max-one", "ACGTAACgt"}, {false, "min-one", "ACGTAacgt"}, {true, "max-one", "ACGTAACgt"}, {true, "min-one", "AcGTAacgt"}}) {
            std::stringstream of;
            std::string alg = t.alg;
            EXPECT_EQ(Optimize(wrapper, kmer_t(0), alg, mpath, of, 3, t.c), 0);
            EXPECT_EQ(of.str(), ">superstring reoptimized=" + t.alg + "\n" + t.want + "\n");
        }
        {
            std::stringstream of;
            std::string alg = "min-run";
            EXPECT_EQ(Optimize(wrapper, kmer_t(0), alg, mpath, of, 3, true), 1);
        }
        auto kMers = wrapper.kh_init_set();
        std::string path = testfa;
        ReadKMersFiltered(kMers, wrapper, kmer_t(0), path, 5, true, 1);
        EXPECT_EQ((size_t)kh_size(kMers), (size_t)11);
        wrapper.kh_destroy_set(kMers);
    }
    {   // Global (tests/global_unittest.h:113-138)
        struct T { std::string want; int k; std::vector<std::string> in; bool c; };
        std::vector<T> tests = {{"TACgt", 3, {"CGT", "TAC", "ACG"}, false}, {"ACGT", 1, {"ACGT"}, false}, {"ACgTtt", 3, {"CGT", "TTT", "ACG"}, false},
                                {"TActTaaGgac", 4, {"TACT", "ACTT", "GGAC", "TAAG"}, false}, {"AtTTgtt", 4, {"ACAA", "ATTT", "AACA"}, true}};
        for (auto &t : tests) {
            std::vector<simplitig_t> v;
            for (auto &s : t.in) v.push_back(simplitig_from_string(s));
            std::stringstream of;
            Global(wrapper, kmer_t(0), v, of, nullptr, t.k, t.c);
            EXPECT_EQ(t.want, of.str());
        }
        std::vector<simplitig_t> empty;
        bool thrown = false;
        std::stringstream of;
        try { Global(wrapper, kmer_t(0), empty, of, nullptr, 3, true); } catch (std::invalid_argument &e) { thrown = std::string(e.what()) == "input cannot be empty"; }
        EXPECT_EQ(thrown, true);
    }
    {   // OverlapHamiltonianPath + SuperstringFromPath (tests/global_unittest.h:11-111)
        std::vector<simplitig_t> v = {simplitig_from_string("ACAA"), simplitig_from_string("ATTT"), simplitig_from_string("AACA")};
        overlapPath got = OverlapHamiltonianPath(wrapper, kmer_t(0), v, 4, true);
        overlapPath want = {{4, 3, 0, 5, (size_t)-1, (size_t)-1}, {2, 2, 3, 3, (unsigned char)-1, (unsigned char)-1}};
        EXPECT_EQ(want.first, got.first);
        EXPECT_EQ(want.second, got.second);
        std::vector<simplitig_t> w = {simplitig_from_string("GCC"), simplitig_from_string("TACG")};
        overlapPath p = {{3, 1, (size_t)-1, (size_t)-1}, {1, 1, (unsigned char)-1, (unsigned char)-1}};
        std::stringstream of, mf;
        size_t L = SuperstringFromPath(wrapper, kmer_t(0), p, w, of, &mf, 3, true);
        EXPECT_EQ(std::string("GcCGta"), of.str());
        EXPECT_EQ((size_t)6, L);
        EXPECT_EQ(mf.str().size(), (size_t)7);
    }
    {   // ReadKMers (tests/parser_unittest.h:38-63) + get_simplitigs (main.cpp:146-170)
        struct T { int k; bool c, cs; size_t want; };
        std::vector<T> tests = {{10, true, false, 3}, {10, true, true, 2}, {5, true, false, 11}, {5, false, false, 11}, {2, true, false, 9}, {2, false, false, 11}};
        for (auto &t : tests) {
            auto kMers = wrapper.kh_init_set();
            std::string path = testfa;
            ReadKMers(kMers, wrapper, kmer_t(0), path, t.k, t.c, t.cs);
            EXPECT_EQ(t.want, (size_t)kh_size(kMers));
            if (!t.cs) {
                auto simplitigs = get_simplitigs(kMers, wrapper, kmer_t(0), t.k, t.c);
                size_t covered = 0;
                for (auto &s : simplitigs) covered += kmers_in_simplitig(s, t.k);
                EXPECT_EQ(t.want, covered);
                std::stringstream of;
                Global(wrapper, kmer_t(0), simplitigs, of, nullptr, t.k, t.c);
                size_t ones = 0;
                for (char ch : of.str()) ones += ch <= 'Z';
                EXPECT_EQ(t.want, ones);
            }
            wrapper.kh_destroy_set(kMers);
        }
    }
    std::cout << (failures ? "FAIL " : "PASS ") << name << std::endl;
}

int main(int argc, char **argv) {
    std::string testfa = argc > 1 ? argv[1] : "tests/golden/test.fa";
    run<kmer64_t, kmer_dict64_t, kh_S64_t>("kmer64_t", testfa);
    run<kmer128_t, kmer_dict128_t, kh_S128_t>("kmer128_t", testfa);
    run<kmer256_t, kmer_dict256_t, kh_S256_t>("kmer256_t", testfa);
    return failures;
}
