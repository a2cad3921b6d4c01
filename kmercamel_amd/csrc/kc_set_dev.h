// Device-side view of the open-addressing k-mer set in HBM (replaces khash's S64/S128/S256, khash.h:215-359).
//
// Layout: `cap` = nblocks * B slots (B = 2^bshift = 4096 slots = one 32 KB region for 64-bit keys), no deletions (the
// simplitig phase marks k-mers as consumed through port links instead of tombstones, so the table stays probe-able
// for the -M mask pass).
//
// Home slot = a function of the k-mer's MIDDLE (k-2)-mer (positions 1..k-2), taken in its canonical orientation:
// block = fastrange(hash of the middle's canonical minimizer (m-mer with the smallest hash over both strands)), offset
// inside the block = hash of the canonical middle.  The 4 candidate neighbours of one side of a k-mer x (x[1..k)+c, resp.
// c+x[0..k-1)) differ only in their outer character, i.e. they share their middle: they share ONE home, so a port of x is
// looked up with one hash, one home and one tag window instead of four.  The tag byte of a slot holds the two outer
// characters of its k-mer (in the middle's canonical orientation) next to 3 hash bits, so the window search tells which of
// the four extensions exist before any key is touched.  Overlapping k-mers share the minimizer of their middles ~90 % of the
// time, so both homes of a k-mer usually lie in its own block (a block can be staged in LDS and probed there).
// Probe sequence: `local_probes` linear steps from the block home, then (overfull block, e.g. a low-complexity
// minimizer) linear probing from a second, minimizer-independent home.  m = 0 (k < 18) falls back to plain hashing of the
// key (tag = 7 hash bits).
// W == 1: the slot is the 64-bit key itself, EMPTY = all ones (never a k-mer: the top bit pair is
// spare for k <= 31), claimed with one 64-bit CAS.  W > 1: a 32-bit state word per slot (EMPTY/BUSY/FULL) is
// claimed by CAS, the key words are then written with agent-scope stores and published by the FULL store.
#pragma once
#include "kc_key.h"

#define KC_ST_EMPTY 0u
#define KC_ST_BUSY 1u
#define KC_ST_FULL 2u

template <int W> struct SetView {
    u64 *keys;
    u32 *state;
    u8 *tags;       // 0 = empty, else 0x80 | a << 5 | b << 3 | 3 hash bits (m != 0; a, b = outer characters) or 0x80 | 7 hash bits (m == 0):
                    // lookups scan 8 slots per 8-byte load and touch a key only on a tag hit
    u64 cap;        // nblocks << bshift
    u64 nblocks;
    int bshift;     // log2(slots per block)
    int m;          // minimizer length (0: plain hashing)
    int k;
    u32 local_probes;
    // hash-sharded tables: this rank holds the global slots [rlo, rlo + rcap) (a whole number of blocks); the three pointers
    // above are biased by -rlo so that they are indexed with GLOBAL slot numbers.  One GPU: rlo = 0, rcap = cap.
    u64 rlo, rcap;
    int complements;   // strand mode of the set (the canonical orientation of a middle is the smaller of it and its reverse complement)
    // dense node ids (phase 1, kc_simp_dev.h): the k-mers of block b are numbered b * nstride + (rank among the block's occupied
    // slots); nstride = the largest number of k-mers in any block (rounded up to a multiple of 8), ncount[b] = k-mers of block b
    // (valid after kcs_prepare_dense; biased like the table pointers on hash-sharded sets).  Ids are dense inside a block and leave
    // nstride - ncount[b] unused ids behind it: what a port is numbered by never needs a directory lookup.
    const u32 *ncount;
    u32 nstride;
};
// Dense ids without a rank directory: once a table is finished (the bucketed build does it on its way, else kcs_prepare_dense), the
// key word 0 of every EMPTY slot holds the number of occupied slots before it in its block (single-word keys: under the mark
// below -- the top pair of a k-mer word is never 11).  The id of the k-mer in slot s is then block * nstride + ann(e) - (e - s) for
// the first empty slot e after s: one more 8-byte load from the line (or the LDS copy) the lookup of s has just touched, instead of
// a second scattered line per lookup.
#define KC_ANN_MARK ((u64)3 << 62)
#define KC_ANN_COUNT(w) ((u64)(w) & 0xffffffffull)
KC_HD bool kc_word_is_empty(u64 w) { return (w >> 62) == 3; }   // single-word keys: ~0 (never annotated) or mark | count

// plain hashing (m == 0): `ph` = key_place_hash(key); its low 7 bits give the tag
KC_HD u8 kc_tag_of(u64 ph) { return (u8)(0x80u | ((u32)ph & 0x7fu)); }
// minimizer blocks (m != 0): outer characters a, b of the k-mer in the canonical orientation of its middle + 3 bits of the
// middle's hash (the offset inside a block is a range reduction of all 32 hash bits, i.e. mostly the high ones)
KC_HD u8 kc_tag_mid(u32 a, u32 b, u64 mh) { return (u8)(0x80u | (a << 5) | (b << 3) | ((u32)mh & 7u)); }
// Homes are confined to the first (block size - local probes - 8) slots of a block: a key's local probe run and the 8-tag
// window of a lookup then never leave the block, so a block can be built or scanned on its own (in LDS).
template <int W> KC_HD u32 set_block_span(const SetView<W> &s) {
    const u32 bs = 1u << s.bshift, tail = s.local_probes + 8u;
    return bs > 2u * tail ? bs - tail : bs / 2u;
}

// hash of a canonical m-mer (f forward, r its reverse complement), 32 bit
KC_HD u32 kc_mmer_hash(u32 f, u32 r) {
    u32 c = f < r ? f : r;
    c *= 0x9E3779B1u;   // one multiply (a bijection; the minimum looks at the well-mixed high bits first, and set_block_home re-mixes
                        // the winner): this runs once per base in the insert kernel and 17x per k-mer in the probe kernel
    return c;
}
// smallest canonical m-mer hash over the m-mers starting at positions i0..i1 (inclusive) of the k-mer x
template <int W> KC_HD u32 key_min_mmh(const Key<W> &x, int k, int m, int i0, int i1) {
    const u32 maskm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1u);
    const int top = 2 * (m - 1);
    u32 f = 0, r = 0;
    for (int j = 0; j < m - 1; ++j) {
        u32 b = key_at(x, k, i0 + j);
        f = (f << 2) | b;
        r = (r >> 2) | ((3u - b) << top);
    }
    u32 best = ~0u;
    for (int i = i0; i <= i1; ++i) {
        u32 b = key_at(x, k, i + m - 1);
        f = ((f << 2) | b) & maskm;
        r = (r >> 2) | ((3u - b) << top);
        u32 h = kc_mmer_hash(f, r);
        best = h < best ? h : best;
    }
    return best;
}
// block home from a minimizer hash and the key hash
template <int W> KC_HD u64 set_block_home(const SetView<W> &s, u32 mmh, u64 keyhash) {
    // mmh is a minimum of ~k-m+1 uniform hashes, i.e. concentrated near 0: re-mix it before the range reduction
    u32 u = mmh * 0x85EBCA6Bu; u ^= u >> 16; u *= 0xC2B2AE35u;   // the range reduction below takes the high bits
    u64 blk = ((u64)u * s.nblocks) >> 32;
    return (blk << s.bshift) | (((u64)(u32)keyhash * set_block_span(s)) >> 32);
}
// second, minimizer-independent home of an overfull block's keys: anywhere in the slot range of the rank that owns the
// block (lookups of a key always run on that rank, so the range is the local one)
template <int W> KC_HD u64 set_global_home(const SetView<W> &s, u64 keyhash) {
    return s.rlo + kc_mulhi64(kc_mix64(keyhash ^ 0x6a09e667f3bcc909ULL), s.rcap);
}
// Canonical middle of a k-mer (k >= 3): M = the smaller of key[1..k-2] and its reverse complement (unidirectional sets: the
// forward one), and the outer characters (a, b) of the k-mer read in M's orientation.  A palindromic middle has two readings
// with the same M: the lexicographically smaller pair is taken, so that the result does not depend on the strand the k-mer
// is given in.
template <int W> KC_HD void key_middle(const Key<W> &key, int k, int complements, Key<W> &M, u32 &a, u32 &b) {
    const Key<W> mf = key_and(key_shr(key, 2), key_lowmask<W>(2 * (k - 2)));
    const u32 c0 = key_at(key, k, 0), c1 = (u32)(key.w[0] & 3);
    M = mf; a = c0; b = c1;
    if (complements) {
        const Key<W> mr = key_rc(mf, k - 2);
        bool rev = key_lt(mr, mf);
        if (key_eq(mr, mf)) rev = (4u * (3u - c1) + (3u - c0)) < (4u * c0 + c1);
        if (rev) { M = mr; a = 3u - c1; b = 3u - c0; }
    }
}
// Where a key lives: home slot, the hash its second home derives from, tag byte.
struct KcPlace { u64 home, kh; u8 tag; };
// m != 0, `mmh` = smallest canonical m-mer hash inside the middle (m-mer starts 1..k-m-1)
template <int W> KC_HD KcPlace set_place_mmh(const SetView<W> &s, const Key<W> &key, u32 mmh) {
    Key<W> M; u32 a, b;
    key_middle(key, s.k, s.complements, M, a, b);
    const u64 mh = key_place_hash(M);
    return KcPlace{set_block_home(s, mmh, mh), mh, kc_tag_mid(a, b, mh)};
}
template <int W> KC_HD KcPlace set_place(const SetView<W> &s, const Key<W> &key) {
    if (s.m == 0) {
        const u64 kh = key_place_hash(key);
        return KcPlace{kc_mulhi64(kc_mix64(kh), s.cap), kh, kc_tag_of(kh)};
    }
    return set_place_mmh(s, key, key_min_mmh(key, s.k, s.m, 1, s.k - s.m - 1));
}
// the part of set_place a block-local build needs (m != 0): offset of the home inside its block, tag
template <int W> KC_HD void set_place_local(const SetView<W> &s, const Key<W> &key, u32 *off, u8 *tag) {
    Key<W> M; u32 a, b;
    key_middle(key, s.k, s.complements, M, a, b);
    const u64 mh = key_place_hash(M);
    *off = (u32)(((u64)(u32)mh * set_block_span(s)) >> 32);
    *tag = kc_tag_mid(a, b, mh);
}
// minimizer length for k (0 = plain hashing): the middle must hold at least one m-mer
static inline int kc_default_mlen(int k) { return k >= 18 ? 16 : 0; }   // 16: an m-mer is exactly one 32-bit word, no masking
// XCD-aware workgroup order for kernels that sweep the table: workgroups are dealt round-robin over the 8 XCDs, so give
// XCD j the j-th contiguous eighth of the sweep -- the workgroups resident on one XCD then work on neighbouring table
// regions and share them through that XCD's L2 (speed only; correctness never depends on placement). Grid = multiple of 8.
#if defined(__HIPCC__)
__device__ __forceinline__ u64 kc_xcd_block_id(int xcd_aware = 1) {
    u64 b = blockIdx.x, chunk = gridDim.x >> 3;
    return xcd_aware ? (b & 7) * chunk + (b >> 3) : b;
}
#endif
// reverse complement of an m-mer held in the low 2m bits
KC_HD u32 kc_mmer_rc(u32 f, int m) { return (u32)(kc_rc_word((u64)f) >> (64 - 2 * m)); }

// i-th slot of the probe sequence that starts at `home`
template <int W> KC_HD u64 set_probe_slot(const SetView<W> &s, u64 home, u64 ghome, u64 i) {
    u64 p = (s.m != 0 && i >= s.local_probes) ? ghome + (i - s.local_probes) : home + i;
    return p >= s.rlo + s.rcap ? p - s.rcap : p;
}

#if defined(__HIPCC__)
// tag of a freshly inserted key; the first 8 tags are mirrored behind the table for set_scan's unaligned 8-byte loads
template <int W> __device__ __forceinline__ void set_write_tag(const SetView<W> &s, u64 slot, u8 tag) {
    s.tags[slot] = tag;
    if (slot - s.rlo < 8) s.tags[s.rcap + slot] = tag;
}


// Insert during the build kernel (concurrent with other inserts). Returns 1 if this call added the key, 0 if it was
// present, -1 if the table is full. *slot_out receives the slot.  `pl` = set_place(s, key) (callers that know the minimizer
// hash of the middle already pass set_place_mmh(...)).
template <int W> __device__ __forceinline__ int set_insert_at(const SetView<W> &s, const Key<W> &key, const KcPlace &pl, u64 *slot_out);
template <int W> __device__ __forceinline__ int set_insert(const SetView<W> &s, const Key<W> &key, u64 *slot_out) {
    return set_insert_at(s, key, set_place(s, key), slot_out);
}
template <int W> __device__ __forceinline__ int set_insert_at(const SetView<W> &s, const Key<W> &key, const KcPlace &pl, u64 *slot_out) {
    const u64 home = pl.home;
    const u64 ghome = set_global_home(s, pl.kh);
    const u64 limit = s.m ? s.rcap + s.local_probes : s.rcap;
    if constexpr (W == 1) {
        unsigned long long *keys = (unsigned long long *)s.keys;
        const unsigned long long kv = key.w[0];
        for (u64 probes = 0; probes < limit; ++probes) {
            u64 slot = set_probe_slot(s, home, ghome, probes);
            unsigned long long cur = keys[slot];
            if (cur == kv) { *slot_out = slot; return 0; }
            if (kc_word_is_empty(cur)) {   // (~0, or the count a finished table left there: kcs_prepare_dense)
                unsigned long long prev = atomicCAS(&keys[slot], cur, kv);
                if (prev == cur) { set_write_tag(s, slot, pl.tag); *slot_out = slot; return 1; }
                if (prev == kv) { *slot_out = slot; return 0; }
            }
        }
        return -1;
    } else {
        u64 probes = 0;
        while (probes < limit) {
            u64 slot = set_probe_slot(s, home, ghome, probes);
            u32 st = __hip_atomic_load(&s.state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st == KC_ST_FULL) {
                // the key loads below must not be hoisted above the state load (the hardware returns vector memory loads in
                // issue order; this keeps the compiler from reordering them)
                asm volatile("" ::: "memory");
                bool eq = true;
                for (int i = 0; i < W; ++i)
                    eq = eq && (__hip_atomic_load(&s.keys[slot * W + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key.w[i]);
                if (eq) { *slot_out = slot; return 0; }
                ++probes;
            } else if (st == KC_ST_EMPTY) {
                u32 prev = atomicCAS(&s.state[slot], KC_ST_EMPTY, KC_ST_BUSY);
                if (prev == KC_ST_EMPTY) {
                    for (int i = 0; i < W; ++i)
                        __hip_atomic_store(&s.keys[slot * W + i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // publish: the key words above are agent-scope (sc1, write-through) stores and every reader uses agent-scope
                    // loads, so it is enough to drain them before the state store.  An agent-scope RELEASE would write back the
                    // whole L2 on every insert (measured: 214 ms per 10^8 k-mers); a workgroup-scope fence compiles to nothing and
                    // lets the state store overtake the keys (seen as a handful of duplicate k-mers) -- hence the explicit wait.
                    // This relies on gfx9 behaviour (stores are counted by vmcnt; sc1 stores are written through): refuse other targets.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "the wide-key publish sequence is written for gfx942 / gfx950 (vmcnt counts stores); port it before building for another target"
#endif
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&s.state[slot], KC_ST_FULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    set_write_tag(s, slot, pl.tag);
                    *slot_out = slot;
                    return 1;
                }
                // lost the race: look at the same slot again
            } else {
                __builtin_amdgcn_s_sleep(1);  // another lane/wave is publishing this slot
            }
        }
        return -1;
    }
}

// Lookup in a finished table (launched after the build kernels completed). Returns the slot or KC_NONE64.
// Tag scan: `count` slots from `start` (wrapping), 8 per (unaligned) 8-byte load. Returns the slot, KC_NONE64 for a proven
// miss (an empty slot was reached), or KC_SCAN_MORE when neither happened within `count` slots.
#define KC_SCAN_MORE (~(u64)1)
template <int W> __device__ __forceinline__ u64 set_scan(const SetView<W> &s, const Key<W> &key, u8 tag, u64 start, u64 count) {
    const u64 pat = 0x0101010101010101ULL * (u64)tag, hi7 = 0x8080808080808080ULL, lo7 = 0x7f7f7f7f7f7f7f7fULL;
    u64 pos = start;
    while (count) {
        // 8 tags from `pos` on in one unaligned load (the 8 tags behind the range mirror its first 8, so no wrap handling here)
        u64 word;
        __builtin_memcpy(&word, s.tags + pos, 8);
        u64 x = word ^ pat;
        u64 zero = ~(((x & lo7) + lo7) | x | lo7);   // 0x80 in every byte of x that is 0  (tag match)
        u64 empty = ~word & hi7;                      // valid tags have bit 7 set
        u64 cand = zero | empty;
        if (count < 8) cand &= ((u64)1 << (8 * count)) - 1;
        while (cand) {
            int b = __builtin_ctzll(cand) >> 3;
            if ((empty >> (8 * b)) & 0x80) return KC_NONE64;
            u64 slot = pos + (u64)b;
            if (slot >= s.rlo + s.rcap) slot -= s.rcap;
            bool eq = true;
            for (int i = 0; i < W; ++i) eq = eq && (s.keys[slot * W + i] == key.w[i]);
            if (eq) return slot;
            cand &= cand - 1;
        }
        if (count <= 8) break;
        count -= 8;
        pos += 8;
        if (pos >= s.rlo + s.rcap) pos -= s.rcap;
    }
    return KC_SCAN_MORE;
}
// `pl` = set_place(s, key)
template <int W> __device__ __forceinline__ u64 set_find_at(const SetView<W> &s, const Key<W> &key, const KcPlace &pl) {
    if (s.m == 0) { u64 r = set_scan(s, key, pl.tag, pl.home, s.rcap); return r == KC_SCAN_MORE ? KC_NONE64 : r; }
    u64 r = set_scan(s, key, pl.tag, pl.home, (u64)s.local_probes);
    if (r != KC_SCAN_MORE) return r;
    r = set_scan(s, key, pl.tag, set_global_home(s, pl.kh), s.rcap);
    return r == KC_SCAN_MORE ? KC_NONE64 : r;
}
template <int W> __device__ __forceinline__ u64 set_find(const SetView<W> &s, const Key<W> &key) { return set_find_at(s, key, set_place(s, key)); }

template <int W> __device__ __forceinline__ bool set_slot_full(const SetView<W> &s, u64 slot) {
    if constexpr (W == 1) return !kc_word_is_empty(s.keys[slot]);
    else return s.state[slot] == KC_ST_FULL;
}
template <int W> __device__ __forceinline__ Key<W> set_key_at(const SetView<W> &s, u64 slot) {
    Key<W> r;
    for (int i = 0; i < W; ++i) r.w[i] = s.keys[slot * W + i];
    return r;
}

// ---- dense node ids <-> slots on a table prepared by kcs_prepare_dense (generic forms: a few dependent loads each; the round-1
// kernels of phase 1 get the same numbers out of the tag windows they have in their hands, kc_simp_dev.h)
// number of occupied slots of block `blk` before its slot offset i (0 <= i <= block size)
template <int W> __device__ __forceinline__ u64 set_count_before(const SetView<W> &s, u64 blk, u32 i) {
    const u32 bs = 1u << s.bshift;
    const u64 base = blk << s.bshift;
    const u64 hi7 = 0x8080808080808080ULL;
    u32 e = i;
    while (e < bs) {   // first empty slot at or after i (8 tags per unaligned load; the 8 bytes behind the table mirror its start)
        u64 wd;
        __builtin_memcpy(&wd, s.tags + base + e, 8);
        u64 emp = ~wd & hi7;
        if (bs - e < 8) emp &= ((u64)1 << (8 * (bs - e))) - 1;
        if (emp) { e += (u32)__builtin_ctzll(emp) >> 3; break; }
        e += 8;
    }
    if (e >= bs) return (u64)s.ncount[blk] - (u64)(bs - i);   // occupied up to the end of the block
    return KC_ANN_COUNT(s.keys[(base + e) * W]) - (u64)(e - i);
}
// id of the k-mer in the (occupied) slot
template <int W> __device__ __forceinline__ u64 set_nid_of_slot(const SetView<W> &s, u64 slot) {
    const u64 blk = slot >> s.bshift;
    return blk * (u64)s.nstride + set_count_before(s, blk, (u32)(slot - (blk << s.bshift)));
}
// is this id in use (ids leave gaps behind every block)?
template <int W> __device__ __forceinline__ bool set_nid_valid(const SetView<W> &s, u64 nid) {
    const u64 blk = nid / s.nstride;
    return (u32)(nid - blk * s.nstride) < s.ncount[blk];
}
// slot of the k-mer with this id (a binary search inside the block: the rare paths only)
template <int W> __device__ __forceinline__ u64 set_slot_of_nid(const SetView<W> &s, u64 nid) {
    const u64 blk = nid / s.nstride, r = nid - blk * s.nstride;
    u32 a = 0, b = (1u << s.bshift) - 1;   // smallest offset i with count_before(i + 1) > r
    while (a < b) { const u32 mid = (a + b) >> 1; if (set_count_before(s, blk, mid + 1) > r) b = mid; else a = mid + 1; }
    return (blk << s.bshift) + a;
}

#endif  // __HIPCC__
