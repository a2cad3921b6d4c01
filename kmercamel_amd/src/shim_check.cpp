// Exercises the drop-in templates exactly the way the reference's driver and unit tests call them
// (main.cpp:146-183; tests/global_unittest.h, tests/simplitigs_unittest.h, tests/parser_unittest.h) for all three key
// widths.  Run on a GPU box: prints PASS/FAIL lines, exit code = number of failures.  (tests/test_gpu_cli.py runs it.)
#include <iostream>
#include <sstream>
#include <algorithm>
#include "global.h"
#include "parser.h"

static int failures = 0;
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) { ++failures; std::cout << "FAIL " << __FILE__ << ":" << __LINE__ << " " #a " != " #b << std::endl; } } while (0)

template <typename kmer_t, typename wrapper_t, typename set_t> void run(const char *name, const std::string &testfa) {
    wrapper_t wrapper;
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
