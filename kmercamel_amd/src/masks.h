// Drop-in for the entry point of the reference's src/masks.h (masks.h:239-261): Optimize for the two data-parallel mask
// objectives, max-one and min-one (OptimizeOnes, masks.h:40-78) -- one kernel pass over the masked superstring against the set
// of its own represented k-mers.  min-run / approx-min-run are a GLPK integer programme and stay outside this build.
#pragma once
#include <string>
#include <iostream>
#include "parser.h"

template <typename kmer_t, typename kh_wrapper_t>
int Optimize(kh_wrapper_t, kmer_t, std::string &algorithm, std::string path, std::ostream &of, int k, bool complements) {
    if (algorithm != "max-one" && algorithm != "min-one") {
        if (algorithm == "min-run" || algorithm == "approx-min-run") std::cerr << "Algorithm '" + algorithm + "' (GLPK ILP) is outside the MI355X build's scope." << std::endl;
        else std::cerr << "Algorithm '" + algorithm + "' not recognized." << std::endl;
        return 1;
    }
    auto recs = kcamd::read_records(path);
    if (recs.empty()) throw std::invalid_argument("couldn't read a masked superstring from " + path);
    const kcamd::Record &ms = recs[0];
    std::string out(ms.seq.size(), 'N');
    int rc = kc_maskopt_ones(kcamd::context(), ms.seq.data(), ms.seq.size(), k, complements, algorithm == "min-one", &out[0]);
    of << ">" << ms.name << " reoptimized=" << algorithm;   // ReprintSequenceHeader (masks.h:27-37)
    if (!ms.comment.empty()) of << " " << ms.comment;
    of << std::endl;
    if (rc == KC_OK || rc == KC_E_INVALID_CHAR) of << out << std::endl;
    if (rc == KC_E_INVALID_CHAR) throw std::invalid_argument("Masked superstring contains invalid characters.");   // masks.h:75-77
    kcamd::check(rc);
    if (recs.size() > 1) throw std::invalid_argument("Expecting only a single FASTA record -- the masked superstring.");
    return 0;
}
