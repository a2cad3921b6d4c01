// Drop-in for the hot-path part of the reference's src/parser.h: AddKMers, ReadKMers, OpenFile semantics, WriteLog,
// WriteName (parser.h:22-49, 88-118, 159-179) with identical signatures; k-mer ingestion runs on the GPU.
#pragma once
#include <string>
#include <iostream>
#include <chrono>
#include <iomanip>
#include <ctime>
#include "kmers.h"
#include "khash_utils.h"
#include "kc_runtime.h"

/// Fill the k-mer dictionary with k-mers from the given sequence (parser.h:22-49).
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
void AddKMers(kh_S_t *kMers, kh_wrapper_t, kmer_t, size_t sequence_length, const char *sequence, int64_t k, bool complements,
              bool case_sensitive = false) {
    kcamd::ensure(kMers, (int)k, complements);
    kcamd::check(kc_set_add_sequence(kMers->h, sequence, sequence_length, case_sensitive ? 1 : 0));
    kcamd::refresh(kMers);
}

/// Load a dictionary of k-mers from a fasta file (parser.h:106-118). Records are handed to the device in one batch,
/// separated by '\n' so that no k-mer spans two records.
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
void ReadKMers(kh_S_t *kMers, kh_wrapper_t wrapper, kmer_t _, std::string &path, int k, bool complements, bool case_sensitive = false) {
    auto recs = kcamd::read_records(path);
    std::string joined;
    size_t total = 0;
    for (auto &r : recs) total += r.seq.size() + 1;
    joined.reserve(total);
    for (auto &r : recs) { joined += r.seq; joined.push_back('\n'); }
    AddKMers(kMers, wrapper, _, joined.size(), joined.data(), k, complements, case_sensitive);
}

/// Load the k-mers that occur at least min_frequency times (parser.h:51-85, 122-141): occurrence counters on the device
/// while the records are added, then the filtered rebuild.
template <typename kmer_t, typename kh_S_t, typename kh_wrapper_t>
void ReadKMersFiltered(kh_S_t *kMers, kh_wrapper_t, kmer_t, std::string &path, int k, bool complements, uint16_t min_frequency) {
    auto recs = kcamd::read_records(path);
    std::string joined;
    size_t total = 0;
    for (auto &r : recs) total += r.seq.size() + 1;
    joined.reserve(total);
    for (auto &r : recs) { joined += r.seq; joined.push_back('\n'); }
    kcamd::ensure(kMers, k, complements);
    kcamd::check(kc_set_enable_counting(kMers->h));
    kcamd::check(kc_set_add_sequence(kMers->h, joined.data(), joined.size(), 0));
    kcamd::check(kc_set_filter_min_frequency(kMers->h, min_frequency < 1 ? 1 : (min_frequency > 255 ? 255 : (int)min_frequency)));
    kcamd::refresh(kMers);
}

inline void WriteLog(const std::string message) {   // parser.h:159-164
    auto snapshot = std::chrono::system_clock::now();
    std::time_t time = std::chrono::system_clock::to_time_t(snapshot);
    std::tm *time_tm = std::localtime(&time);
    std::cerr << "[" << std::put_time(time_tm, "%H:%M:%S") << "] " << message << std::endl;
}

/// Print the fasta file header (parser.h:167-179).
inline void WriteName(const std::string &dataset, const std::string &algorithm, const int k, const bool maxone, const bool unidirectional,
                      std::ostream &of) {
    of << ">maskedsuperstring dataset='" << dataset << "' k=" << k << " alg=" << algorithm << " mask=" << (maxone ? "max-one " : "min-one ")
       << "mode=" << (unidirectional ? "unidirectional" : "bidirectional") << std::endl;
}
