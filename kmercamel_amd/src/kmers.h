// Drop-in for the reference's src/kmers.h (host-side helpers only; the hot-path arithmetic lives in csrc/kc_key.h).
// Same names and meaning: kmer64_t/kmer128_t/kmer256_t, nucleotideToInt, letters, BitPrefix, BitSuffix,
// ReverseComplement, AtIndex, NucleotideAtIndex, NumberToKMer, Masked  (kmers.h:11-127).
#pragma once
#include <cstdint>
#include <string>

// minimal 256-bit unsigned for k in [64,127] (the reference vendors calccrypto's uint256_t; only what callers of this
// interface need is provided)
struct uint256_t {
    __uint128_t lo = 0, hi = 0;
    uint256_t() = default;
    uint256_t(uint64_t v) : lo(v), hi(0) {}
    uint256_t(int v) : lo((__uint128_t)(__int128)v), hi(v < 0 ? ~(__uint128_t)0 : 0) {}
    uint256_t(__uint128_t h, __uint128_t l) : lo(l), hi(h) {}
    __uint128_t lower() const { return lo; }
    __uint128_t upper() const { return hi; }
    explicit operator uint64_t() const { return (uint64_t)lo; }
    explicit operator __uint128_t() const { return lo; }
    uint256_t operator<<(int s) const {
        if (s <= 0) return *this;
        if (s >= 256) return uint256_t();
        if (s >= 128) return uint256_t(lo << (s - 128), 0);
        return uint256_t((hi << s) | (lo >> (128 - s)), lo << s);
    }
    uint256_t operator>>(int s) const {
        if (s <= 0) return *this;
        if (s >= 256) return uint256_t();
        if (s >= 128) return uint256_t(0, hi >> (s - 128));
        return uint256_t(hi >> s, (lo >> s) | (hi << (128 - s)));
    }
    uint256_t operator|(const uint256_t &o) const { return uint256_t(hi | o.hi, lo | o.lo); }
    uint256_t operator&(const uint256_t &o) const { return uint256_t(hi & o.hi, lo & o.lo); }
    uint256_t operator^(const uint256_t &o) const { return uint256_t(hi ^ o.hi, lo ^ o.lo); }
    uint256_t operator~() const { return uint256_t(~hi, ~lo); }
    uint256_t operator-(const uint256_t &o) const { uint256_t r; r.lo = lo - o.lo; r.hi = hi - o.hi - (lo < o.lo ? 1 : 0); return r; }
    uint256_t &operator<<=(int s) { *this = *this << s; return *this; }
    uint256_t &operator>>=(int s) { *this = *this >> s; return *this; }
    uint256_t &operator|=(const uint256_t &o) { *this = *this | o; return *this; }
    bool operator==(const uint256_t &o) const { return lo == o.lo && hi == o.hi; }
    bool operator!=(const uint256_t &o) const { return !(*this == o); }
    bool operator<(const uint256_t &o) const { return hi < o.hi || (hi == o.hi && lo < o.lo); }
    bool operator<=(const uint256_t &o) const { return !(o < *this); }
};

typedef __uint128_t kmer128_t;
typedef uint64_t kmer64_t;
typedef uint256_t kmer256_t;

inline uint8_t kc_nucleotide_code(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
struct kc_nucleotide_table { uint8_t operator[](size_t c) const { return kc_nucleotide_code((unsigned char)c); } };
static const kc_nucleotide_table nucleotideToInt{};
const char letters[4]{'A', 'C', 'G', 'T'};

namespace kcamd {
inline uint64_t low64(uint64_t x) { return x; }
inline uint64_t low64(__uint128_t x) { return (uint64_t)x; }
inline uint64_t low64(const uint256_t &x) { return (uint64_t)x.lo; }
inline void to_words(uint64_t x, uint64_t *w) { w[0] = x; }
inline void to_words(__uint128_t x, uint64_t *w) { w[0] = (uint64_t)x; w[1] = (uint64_t)(x >> 64); }
inline void to_words(const uint256_t &x, uint64_t *w) { w[0] = (uint64_t)x.lo; w[1] = (uint64_t)(x.lo >> 64); w[2] = (uint64_t)x.hi; w[3] = (uint64_t)(x.hi >> 64); }
inline void from_words(const uint64_t *w, uint64_t &x) { x = w[0]; }
inline void from_words(const uint64_t *w, __uint128_t &x) { x = ((__uint128_t)w[1] << 64) | w[0]; }
inline void from_words(const uint64_t *w, uint256_t &x) { x = uint256_t(((__uint128_t)w[3] << 64) | w[2], ((__uint128_t)w[1] << 64) | w[0]); }
template <typename T> struct words_of { static const int value = sizeof(T) / 8; };
}  // namespace kcamd

template <typename kmer_t> kmer_t BitPrefix(kmer_t kMer, int k, int d) { return kMer >> (2 * (k - d)); }
template <typename kmer_t> kmer_t BitSuffix(kmer_t kMer, int d) { return kMer & ((kmer_t(1) << (2 * d)) - kmer_t(1)); }
template <typename kmer_t> kmer_t ReverseComplement(kmer_t kMer, int k) {
    kmer_t r = kmer_t(0);
    for (int i = 0; i < k; ++i) { r = (r << 2) | kmer_t((uint64_t)(3 - (kcamd::low64(kMer) & 3))); kMer = kMer >> 2; }
    return r;
}
template <typename kmer_t> inline kmer_t AtIndex(kmer_t encoded, int k, int index) { return (encoded >> (2 * (k - index - 1))) & kmer_t(3); }
template <typename kmer_t> inline char NucleotideAtIndex(kmer_t encoded, int k, int index) { return letters[kcamd::low64(AtIndex(encoded, k, index))]; }
template <typename kmer_t> std::string NumberToKMer(kmer_t encoded, int length) {
    std::string ret(length, 'N');
    for (int i = 0; i < length; ++i) { ret[length - i - 1] = letters[kcamd::low64(encoded) & 3]; encoded = encoded >> 2; }
    return ret;
}
inline char Masked(char c, bool mask) {
    bool upper = c <= 'Z';
    if (mask) return upper ? c : (char)(c - ('a' - 'A'));
    return upper ? (char)(c + ('a' - 'A')) : c;
}
