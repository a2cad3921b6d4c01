// k-mer key arithmetic for 64/128/256-bit keys as W little-endian 64-bit words (W = 1, 2, 4).
// Compiles for gfx950 device code (hipcc) and for the host (hipcc host pass or plain g++; the latter is used by
// the CPU simulation harness in tests/sim to exercise the per-bucket logic without a GPU).
//
// Semantics follow the reference's k-mer layer (/root/reference/src/kmers.h): 2 bits per nucleotide, A=0 C=1 G=2 T=3
// (kmers.h:15-32), first nucleotide in the most significant used bits, BitPrefix/BitSuffix (kmers.h:36-44),
// ReverseComplement (kmers.h:93-95).  The reverse complement is written from scratch for this ISA:
// one v_bfrev per 32 bits + a pair swap + complement + a funnel shift.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KC_HD __host__ __device__ __forceinline__
#else
#define KC_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define KC_NONE64 (~(u64)0)

template <int W> struct Key { u64 w[W]; };

template <int W> KC_HD Key<W> key_zero() { Key<W> r; for (int i = 0; i < W; ++i) r.w[i] = 0; return r; }
template <int W> KC_HD Key<W> key_ones() { Key<W> r; for (int i = 0; i < W; ++i) r.w[i] = ~(u64)0; return r; }
template <int W> KC_HD bool key_eq(const Key<W> &a, const Key<W> &b) {
    bool e = true;
    for (int i = 0; i < W; ++i) e = e && (a.w[i] == b.w[i]);
    return e;
}
template <int W> KC_HD bool key_lt(const Key<W> &a, const Key<W> &b) {
    for (int i = W - 1; i >= 0; --i) {
        if (a.w[i] < b.w[i]) return true;
        if (a.w[i] > b.w[i]) return false;
    }
    return false;
}
template <int W> KC_HD bool key_is_ones(const Key<W> &a) {
    bool e = true;
    for (int i = 0; i < W; ++i) e = e && (a.w[i] == ~(u64)0);
    return e;
}
// logical shifts by 0 <= s < 64*W.  All word indices are compile-time constants (a barrel shifter over the words): a
// runtime-indexed register array would be placed in scratch memory by hipcc.
template <int W> KC_HD Key<W> key_shr(const Key<W> &a, int s) {
    if constexpr (W == 1) { Key<W> r; r.w[0] = a.w[0] >> s; return r; }
    Key<W> t = a;
    const int ws = s >> 6, bs = s & 63;
    for (int step = 1; step < W; step <<= 1) {
        const bool on = (ws & step) != 0;
        Key<W> u;
        for (int i = 0; i < W; ++i) u.w[i] = (i + step < W) ? t.w[i + step] : 0;
        for (int i = 0; i < W; ++i) t.w[i] = on ? u.w[i] : t.w[i];
    }
    Key<W> r;
    for (int i = 0; i < W; ++i) {
        u64 hi = (i + 1 < W) ? t.w[i + 1] : 0;
        r.w[i] = bs ? ((t.w[i] >> bs) | (hi << (64 - bs))) : t.w[i];
    }
    return r;
}
template <int W> KC_HD Key<W> key_shl(const Key<W> &a, int s) {
    if constexpr (W == 1) { Key<W> r; r.w[0] = a.w[0] << s; return r; }
    Key<W> t = a;
    const int ws = s >> 6, bs = s & 63;
    for (int step = 1; step < W; step <<= 1) {
        const bool on = (ws & step) != 0;
        Key<W> u;
        for (int i = 0; i < W; ++i) u.w[i] = (i - step >= 0) ? t.w[i - step] : 0;
        for (int i = 0; i < W; ++i) t.w[i] = on ? u.w[i] : t.w[i];
    }
    Key<W> r;
    for (int i = W - 1; i >= 0; --i) {
        u64 lo = (i - 1 >= 0) ? t.w[i - 1] : 0;
        r.w[i] = bs ? ((t.w[i] << bs) | (lo >> (64 - bs))) : t.w[i];
    }
    return r;
}
// low `bits` bits set (0 <= bits <= 64*W)
template <int W> KC_HD Key<W> key_lowmask(int bits) {
    Key<W> r;
    for (int i = 0; i < W; ++i) {
        int b = bits - 64 * i;
        r.w[i] = b <= 0 ? 0 : (b >= 64 ? ~(u64)0 : (((u64)1 << b) - 1));
    }
    return r;
}
template <int W> KC_HD Key<W> key_and(const Key<W> &a, const Key<W> &b) { Key<W> r; for (int i = 0; i < W; ++i) r.w[i] = a.w[i] & b.w[i]; return r; }
template <int W> KC_HD Key<W> key_or(const Key<W> &a, const Key<W> &b) { Key<W> r; for (int i = 0; i < W; ++i) r.w[i] = a.w[i] | b.w[i]; return r; }

// ((x << 2) & mask_k) | c      -- append nucleotide c on the right (parser.h:38, simplitigs.h:112,115)
template <int W> KC_HD Key<W> key_append(const Key<W> &x, u64 c, int k) {
    Key<W> r = key_and(key_shl(x, 2), key_lowmask<W>(2 * k));
    r.w[0] |= c;
    return r;
}
// (x >> 2) | (c << 2(k-1))     -- prepend nucleotide c on the left (parser.h:39, simplitigs.h:113,116)
template <int W> KC_HD Key<W> key_prepend(const Key<W> &x, u64 c, int k) {
    Key<W> r = key_shr(x, 2);
    const int s = 2 * (k - 1), ws = s >> 6;
    const u64 bits = c << (s & 63);
    for (int i = 0; i < W; ++i) r.w[i] |= (ws == i) ? bits : 0;
    return r;
}
// BitPrefix(x, k, d) = x >> 2(k-d); BitSuffix(x, d) = x & (4^d - 1)    (kmers.h:36-44)
template <int W> KC_HD Key<W> key_prefix(const Key<W> &x, int k, int d) { return key_shr(x, 2 * (k - d)); }
template <int W> KC_HD Key<W> key_suffix(const Key<W> &x, int d) { return key_and(x, key_lowmask<W>(2 * d)); }
// nucleotide at index i (0 = first)   (kmers.h:99-102)
template <int W> KC_HD u32 key_at(const Key<W> &x, int k, int i) {
    const int s = 2 * (k - 1 - i), ws = s >> 6;
    u64 v = x.w[0];
    for (int j = 1; j < W; ++j) v = (ws == j) ? x.w[j] : v;
    return (u32)((v >> (s & 63)) & 3);
}

KC_HD u64 kc_brev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    x = ((x >> 8) & 0x00ff00ff00ff00ffULL) | ((x & 0x00ff00ff00ff00ffULL) << 8);
    x = ((x >> 16) & 0x0000ffff0000ffffULL) | ((x & 0x0000ffff0000ffffULL) << 16);
    return (x >> 32) | (x << 32);
#endif
}
// reverse the 32 2-bit groups of a word and complement them
KC_HD u64 kc_rc_word(u64 x) {
    u64 r = kc_brev64(x);                                                            // bit reverse
    r = ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);     // restore bit order inside each pair
    return ~r;
}
// ReverseComplement(x, k)  (kmers.h:93-95)
template <int W> KC_HD Key<W> key_rc(const Key<W> &x, int k) {
    Key<W> r;
    for (int i = 0; i < W; ++i) r.w[i] = kc_rc_word(x.w[W - 1 - i]);
    return key_shr(r, 64 * W - 2 * k);
}

// Layout-free 64-bit key hash (murmur3 finaliser folded over the words, most significant first). The same function
// is restated in oracle/kc_oracle.cpp (key_hash) because the canonical-order simplitig rule is defined with it.
KC_HD u64 kc_mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
template <int W> KC_HD u64 key_hash(const Key<W> &x) {
    u64 h = kc_mix64(x.w[W - 1]);
    for (int i = W - 2; i >= 0; --i) h = kc_mix64(h ^ x.w[i]);
    return h;
}
// Placement hash of the k-mer set: 32 bits, two 32-bit multiplies per key word (the 64-bit murmur finaliser above costs
// ~4x as many multiply issues, and the neighbour-probe kernel hashes 8 candidates per k-mer).  Only the LAYOUT of the
// table depends on it -- results never do (the canonical simplitig rule uses key_hash) -- so it is free to be cheap.
template <int W> KC_HD u64 key_place_hash(const Key<W> &x) {
    u32 h = 0;
    for (int i = W - 1; i >= 0; --i) {
        h = (h ^ (u32)x.w[i]) * 0x9E3779B1u; h ^= h >> 15;
        h = (h ^ (u32)(x.w[i] >> 32)) * 0x85EBCA77u; h ^= h >> 13;
    }
    return (u64)h;
}
KC_HD u64 kc_port_term(u64 keyhash, int port) {
    return kc_mix64(keyhash ^ (port ? 0x9e3779b97f4a7c15ULL : 0x7f4a7c159e3779b9ULL));
}

KC_HD u64 kc_mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// words needed for a k-mer: k<32 -> 1, k<64 -> 2, else 4   (main.cpp:309-315)
static inline int kc_words_for_k(int k) { return k < 32 ? 1 : (k < 64 ? 2 : 4); }
