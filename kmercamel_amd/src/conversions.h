// Drop-in for the reference's src/conversions.h (split_ms :16-33, join_ms :35-43, ms_to_spss :45-72, spss_to_ms :74-91):
// the representation converters that sit either side of the hot path (ms <-> superstring+mask, ms <-> (r)SPSS).
// They are byte shuffles over a file that is read once, so they stay on the host; each one works on whole buffers and
// writes its output with a single stream write instead of the reference's character-at-a-time inserts.
#pragma once
#include <cstdint>
#include <iostream>
#include <string>
#include <vector>
#include "kmers.h"
#include "kc_runtime.h"

inline bool is_upper(char c) { return c <= 'Z'; }                               // conversions.h:6-8
inline char to_upper(char c) { return is_upper(c) ? c : (char)(c - ('a' - 'A')); }   // conversions.h:10-14

namespace kcamd {
// Only the first record matters to split_ms / ms_to_spss (a single kseq_read); a file without any record gives an
// empty sequence, which is what the reference's loop over a negative length amounts to.
inline std::string first_sequence(const std::string &path) {
    auto recs = read_records(path);
    return recs.empty() ? std::string() : std::move(recs[0].seq);
}
}  // namespace kcamd

/// ms -> superstring + mask, one line each (conversions.h:16-33).
inline void split_ms(std::ostream &superstring, std::ostream &mask, std::string path) {
    std::string ms = kcamd::first_sequence(path);
    std::string bits(ms.size(), '0');
    // (branch-free, so that the compiler vectorises it: masks flip case at random places, and a mispredicted branch per character made this
    // loop the slowest part of the command -- 1.7 of 2.9 s per 300 MB)
    char *c = &ms[0], *b = &bits[0];
    for (size_t i = 0, n = ms.size(); i < n; ++i) {
        const int lower = c[i] > 'Z';
        b[i] = (char)('1' - lower);
        c[i] = (char)(c[i] - (lower << 5));
    }
    // the reference interleaves the two streams per character and ends the mask line first; when both point at the
    // same stream (stdout) the order of the two whole lines below is the only part of that interleaving kept.
    mask << bits << std::endl;
    superstring << ms << std::endl;
}

/// superstring + mask (whitespace-delimited tokens) -> ms (conversions.h:35-43). A mask shorter than the superstring
/// counts as '0' from its end on (the reference reads past the end of the string there).
inline void join_ms(std::istream &superstringf, std::istream &maskf, std::ostream &of) {
    std::string s, mask;
    superstringf >> s;
    maskf >> mask;
    {
        const size_t m = mask.size() < s.size() ? mask.size() : s.size();
        char *c = &s[0];
        const char *b = mask.data();
        for (size_t i = 0; i < m; ++i) c[i] = Masked(c[i], b[i] == '1');
        for (size_t i = m; i < s.size(); ++i) c[i] = Masked(c[i], false);
    }
    of << ">superstring" << std::endl << s << std::endl;
}

/// ms -> rSPSS: every maximal run of ones, extended by the k-1 characters that follow it (conversions.h:45-72).
inline void ms_to_spss(std::string path, std::ostream &of, int k) {
    const std::string ms = kcamd::first_sequence(path);
    const size_t n = ms.size();
    std::string out;
    out.reserve(n + n / 8);
    size_t counter = 0, i = 0;
    while (i < n) {
        if (!is_upper(ms[i])) { ++i; continue; }
        size_t e = i;
        while (e < n && is_upper(ms[e])) ++e;
        out += '>';
        out += std::to_string(counter++);
        out += '\n';
        out.append(ms, i, e - i);
        if (e < n) {   // a run that reaches the end of the string gets neither its tail nor a line end in the reference
            size_t tail = std::min(n, e + (size_t)(k > 1 ? k - 1 : 0));
            for (size_t j = e; j < tail; ++j) out += to_upper(ms[j]);
            out += '\n';
        }
        i = e + (e < n ? 1 : 0);
    }
    of << out;
}

/// (r)SPSS -> ms: records of at least k characters concatenated, ones at every k-mer start (conversions.h:74-91).
inline void spss_to_ms(std::string path, std::ostream &of, int k) {
    auto recs = kcamd::read_records(path);
    size_t total = 0;
    for (auto &r : recs) total += r.seq.size();
    std::string out;
    out.reserve(total);
    for (auto &r : recs) {
        const int64_t l = (int64_t)r.seq.size();
        if (l < k) continue;
        for (int64_t i = 0; i < l; ++i) out += Masked(r.seq[i], i <= l - k);
    }
    of << ">superstring " << path << std::endl << out << std::endl;
}
