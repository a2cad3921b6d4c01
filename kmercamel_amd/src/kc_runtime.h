// Host runtime shared by the drop-in headers: one lazily created kc_ctx per process (GPU chosen by KMERCAMEL_GPU,
// default 0), C-ABI error codes turned back into the exception types the reference throws, and the record reader.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <mutex>
#include <thread>
#include <zlib.h>
#include <sys/stat.h>
#include "../../include/kmercamel_amd.h"

namespace kcamd {

inline kc_ctx *context() {
    static std::once_flag once;
    static kc_ctx *ctx = nullptr;
    static int rc = KC_OK;
    std::call_once(once, [] {
        const char *g = std::getenv("KMERCAMEL_GPU");
        rc = kc_ctx_create(g ? std::atoi(g) : 0, &ctx);
    });
    if (rc != KC_OK) throw std::runtime_error("kmercamel_amd: no usable MI355X / HIP device (kc_ctx_create returned " + std::to_string(rc) + ")");
    return ctx;
}
// Device and context creation takes ~0.1-0.2 s: the CLI starts it in the background and reads the input meanwhile.
inline void warm_context() {
    std::thread([] { try { (void)context(); } catch (...) {} }).detach();
}

// C codes -> the reference's exceptions (global.h:219-221, parser.h:95-97, simplitigs.h:82, masks.h:75-77)
inline void check(int rc) {
    if (rc == KC_OK) return;
    std::string msg = kc_last_error(context());
    switch (rc) {
        case KC_E_EMPTY: throw std::invalid_argument("input cannot be empty");
        case KC_E_INVALID_CHAR: throw std::invalid_argument(msg.empty() ? "invalid character in input" : msg);
        case KC_E_IO: throw std::invalid_argument(msg);
        case KC_E_ARG: throw std::invalid_argument(msg.empty() ? "bad argument" : msg);
        default: throw std::runtime_error("kmercamel_amd: " + msg + " (code " + std::to_string(rc) + ")");
    }
}

struct Record { std::string name, comment, seq; };

// Whole file (plain or gz, "-" = stdin) into memory; throws like OpenFile (parser.h:88-101).
inline std::string slurp(const std::string &path) {
    gzFile f;
    if (path == "-") f = gzdopen(0, "r");
    else {
        FILE *in = fopen(path.c_str(), "r");
        if (in == nullptr) throw std::invalid_argument("couldn't open file " + path);
        f = gzdopen(fileno(in), "r");
    }
    std::string buf;
    std::vector<char> tmp(1 << 20);
    int n;
    while ((n = gzread(f, tmp.data(), (unsigned)tmp.size())) > 0) buf.append(tmp.data(), (size_t)n);
    gzclose(f);
    return buf;
}

// Plain (non-gzip) regular files are read with one read into a buffer of the file's size; everything else goes through zlib.
inline std::string slurp_fast(const std::string &path) {
    if (path != "-") {
        FILE *in = fopen(path.c_str(), "rb");
        if (in == nullptr) throw std::invalid_argument("couldn't open file " + path);
        struct stat sti;   // only a regular file can be sized and read in one piece; a FIFO must not even be peeked at (it cannot be re-opened)
        const bool regular = fstat(fileno(in), &sti) == 0 && S_ISREG(sti.st_mode);
        if (!regular) {
            gzFile f = gzdopen(fileno(in), "r");
            std::string buf;
            std::vector<char> tmp(1 << 20);
            int n;
            while ((n = gzread(f, tmp.data(), (unsigned)tmp.size())) > 0) buf.append(tmp.data(), (size_t)n);
            gzclose(f);
            return buf;
        }
        unsigned char magic[2] = {0, 0};
        size_t got = fread(magic, 1, 2, in);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) && fseek(in, 0, SEEK_END) == 0) {
            long size = ftell(in);
            if (size >= 0 && fseek(in, 0, SEEK_SET) == 0) {
                std::string buf((size_t)size, '\0');
                size_t n = size ? fread(&buf[0], 1, (size_t)size, in) : 0;
                fclose(in);
                buf.resize(n);
                return buf;
            }
        }
        fclose(in);
    }
    return slurp(path);
}

// FASTA fast path of the tokenizer below, straight into the form the device wants: the record sequences joined by '\n'
// (or, for -S, concatenated with their offsets).  Same rules (kseq.h:183-224): everything before the first '>' / '@' is
// skipped, a line starting with '>' or '@' starts a record, empty lines are skipped, a '\r' ending the accumulated
// sequence is dropped.  Returns false as soon as a '+' line shows that the file is FASTQ (the general parser takes over).
inline bool join_fasta(const std::string &b, bool concatenate, std::string &joined, std::vector<uint64_t> &offsets, uint64_t &n_records) {
    const char *p = b.data(), *end = p + b.size();
    joined.clear(); offsets.clear(); n_records = 0;
    joined.reserve(b.size() + 16);
    while (p < end && *p != '>' && *p != '@') ++p;
    while (p < end) {
        // p is at a header character (a header character that is the last byte of the file starts no record): skip the line
        if (p + 1 >= end) break;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = nl ? nl + 1 : end;
        const size_t start = joined.size();
        if (concatenate) offsets.push_back(start);
        bool next_header = false;
        while (p < end) {
            const char c = *p;
            if (c == '>' || c == '@') { next_header = true; break; }
            if (c == '+') return false;
            if (c == '\n') { ++p; continue; }
            nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            joined.append(p, (size_t)(le - p));
            if (joined.size() - start > 1 && joined.back() == '\r') joined.pop_back();
            p = nl ? nl + 1 : end;
        }
        ++n_records;
        if (!concatenate) joined.push_back('\n');
        if (!next_header) break;
    }
    if (concatenate) offsets.push_back(joined.size());
    return true;
}

// Records with the semantics of the reference's tokenizer (kseq.h:183-224): multi-line FASTA and FASTQ, header char
// searched anywhere after a FASTQ record, one trailing '\r' stripped per line, truncated FASTQ ends the stream.
inline std::vector<Record> parse_records(const std::string &b) {
    std::vector<Record> out;
    size_t p = 0, n = b.size();
    int last_char = 0;
    auto take_line = [&](std::string &dst) {
        size_t e = b.find('\n', p);
        if (e == std::string::npos) e = n;
        dst.append(b, p, e - p);
        if (dst.size() > 1 && dst.back() == '\r') dst.pop_back();
        p = e < n ? e + 1 : n;
    };
    while (true) {
        if (last_char == 0) {
            while (p < n && b[p] != '>' && b[p] != '@') ++p;
            if (p >= n) break;
            last_char = b[p++];
        }
        if (p >= n) break;
        Record r;
        size_t q = p;
        while (q < n && !isspace((unsigned char)b[q])) ++q;
        r.name = b.substr(p, q - p);
        if (q < n) {
            int delim = b[q];
            p = q + 1;
            if (delim != '\n') take_line(r.comment);
        } else p = n;
        int c = -1;
        while (p < n) {
            c = (unsigned char)b[p++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') { c = -1; continue; }
            r.seq.push_back((char)c);
            take_line(r.seq);
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') {
            out.push_back(std::move(r));
            if (c == -1) break;
            continue;
        }
        size_t e = b.find('\n', p);
        if (e == std::string::npos) break;
        p = e + 1;
        std::string qual;
        while (p < n) {
            take_line(qual);
            if (!(qual.size() < r.seq.size())) break;
        }
        last_char = 0;
        if (qual.size() != r.seq.size()) break;
        out.push_back(std::move(r));
    }
    return out;
}

inline std::vector<Record> read_records(const std::string &path) { return parse_records(slurp(path)); }

}  // namespace kcamd
