// join_fasta (the CLI's one-pass FASTA loader) against the general kseq-semantics tokenizer parse_records, on edge cases and
// random byte soups.  Host-only: g++ tests/host/parse_check.cpp -Ikmercamel_amd/src -lz
#include <cstdio>
#include <random>
#include "kc_runtime.h"

static bool same(const std::string &b, bool concatenate) {
    std::string joined; std::vector<uint64_t> offsets; uint64_t n = 0;
    const bool fast = kcamd::join_fasta(b, concatenate, joined, offsets, n);
    auto recs = kcamd::parse_records(b);
    if (!fast) return true;   // FASTQ-like input: the CLI falls back to parse_records itself
    std::string want; std::vector<uint64_t> woff;
    for (auto &r : recs) { if (concatenate) woff.push_back(want.size()); want += r.seq; if (!concatenate) want.push_back('\n'); }
    if (concatenate) woff.push_back(want.size());
    return n == recs.size() && joined == want && (!concatenate || offsets == woff);
}

int main() {
    const char *cases[] = {"", ">", ">a", ">a\n", ">a\nACGT", ">a\nACGT\n", ">a\nAC\nGT\n>b\nTT\n", "junk>a\nAC\n", ">a\r\nAC\r\nGT\r\n>b\r\n\r\nA\r\n",
                           ">a\n\n\nAC\n\n>b\n", ">a\n>b\n>c\nA", ">a\nAC\n@b\nGG\n", "@a\nAC\n", ">a desc more\nAC>GT\nA@C\n", ">a\nAC\n+\nII\n",
                           "@r\nACGT\n+\nIIII\n", ">a\n\r\n", ">a\nA\r", ">a\n\rA\n", "\n\n>a\nAC", ">a\tcomment\nAC\n>", ">a\nAC\n@", "no header at all\nACGT\n"};
    int bad = 0;
    for (const char *c : cases)
        for (int conc = 0; conc < 2; ++conc)
            if (!same(c, conc)) { printf("MISMATCH on case \"%s\" (concatenate=%d)\n", c, conc); ++bad; }
    std::mt19937 rng(12345);
    const char alphabet[] = ">@+\n\rACGTN acgt\t";
    for (int it = 0; it < 200000 && !bad; ++it) {
        std::string b;
        int len = rng() % 60;
        for (int i = 0; i < len; ++i) b.push_back(alphabet[rng() % (sizeof(alphabet) - 1)]);
        for (int conc = 0; conc < 2; ++conc)
            if (!same(b, conc)) { printf("MISMATCH on random case %d (concatenate=%d)\n", it, conc); ++bad; }
    }
    printf(bad ? "parse_check FAILED\n" : "parse_check ok\n");
    return bad ? 1 : 0;
}
