// Device-side view of the open-addressing k-mer set in HBM (replaces khash's S64/S128/S256, khash.h:215-359).
//
// Layout: `cap` slots, linear probing from fastrange(hash) (slot = mulhi(hash, cap)), no deletions (the simplitig
// phase marks k-mers as consumed through port links instead of tombstones, so the table stays probe-able for the
// -M mask pass).  W == 1: the slot is the 64-bit key itself, EMPTY = all ones (never a k-mer: the top bit pair is
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
    u64 cap;
};

#if defined(__HIPCC__)

template <int W> __device__ __forceinline__ u64 set_home(const SetView<W> &s, const Key<W> &key) {
    return kc_mulhi64(key_hash(key), s.cap);
}

// Insert during the build kernel (concurrent with other inserts). Returns 1 if this call added the key, 0 if it was
// present, -1 if the table is full. *slot_out receives the slot.
template <int W> __device__ __forceinline__ int set_insert(const SetView<W> &s, const Key<W> &key, u64 *slot_out) {
    u64 slot = set_home(s, key);
    if constexpr (W == 1) {
        unsigned long long *keys = (unsigned long long *)s.keys;
        const unsigned long long kv = key.w[0];
        for (u64 probes = 0; probes < s.cap; ++probes) {
            unsigned long long cur = keys[slot];
            if (cur == kv) { *slot_out = slot; return 0; }
            if (cur == ~0ULL) {
                unsigned long long prev = atomicCAS(&keys[slot], ~0ULL, kv);
                if (prev == ~0ULL) { *slot_out = slot; return 1; }
                if (prev == kv) { *slot_out = slot; return 0; }
            }
            slot = (slot + 1 == s.cap) ? 0 : slot + 1;
        }
        return -1;
    } else {
        u64 probes = 0;
        while (probes < s.cap) {
            u32 st = __hip_atomic_load(&s.state[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (st == KC_ST_FULL) {
                bool eq = true;
                for (int i = 0; i < W; ++i)
                    eq = eq && (__hip_atomic_load(&s.keys[slot * W + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key.w[i]);
                if (eq) { *slot_out = slot; return 0; }
                slot = (slot + 1 == s.cap) ? 0 : slot + 1;
                ++probes;
            } else if (st == KC_ST_EMPTY) {
                u32 prev = atomicCAS(&s.state[slot], KC_ST_EMPTY, KC_ST_BUSY);
                if (prev == KC_ST_EMPTY) {
                    for (int i = 0; i < W; ++i)
                        __hip_atomic_store(&s.keys[slot * W + i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&s.state[slot], KC_ST_FULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
template <int W> __device__ __forceinline__ u64 set_find(const SetView<W> &s, const Key<W> &key) {
    u64 slot = set_home(s, key);
    if constexpr (W == 1) {
        const u64 kv = key.w[0];
        for (u64 probes = 0; probes < s.cap; ++probes) {
            u64 cur = s.keys[slot];
            if (cur == kv) return slot;
            if (cur == ~(u64)0) return KC_NONE64;
            slot = (slot + 1 == s.cap) ? 0 : slot + 1;
        }
        return KC_NONE64;
    } else {
        for (u64 probes = 0; probes < s.cap; ++probes) {
            u32 st = s.state[slot];
            if (st == KC_ST_EMPTY) return KC_NONE64;
            bool eq = true;
            for (int i = 0; i < W; ++i) eq = eq && (s.keys[slot * W + i] == key.w[i]);
            if (eq) return slot;
            slot = (slot + 1 == s.cap) ? 0 : slot + 1;
        }
        return KC_NONE64;
    }
}

template <int W> __device__ __forceinline__ bool set_slot_full(const SetView<W> &s, u64 slot) {
    if constexpr (W == 1) return s.keys[slot] != ~(u64)0;
    else return s.state[slot] == KC_ST_FULL;
}
template <int W> __device__ __forceinline__ Key<W> set_key_at(const SetView<W> &s, u64 slot) {
    Key<W> r;
    for (int i = 0; i < W; ++i) r.w[i] = s.keys[slot * W + i];
    return r;
}

#endif  // __HIPCC__
