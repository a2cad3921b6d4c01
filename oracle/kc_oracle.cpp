// TEST INFRASTRUCTURE ONLY -- the CPU oracle.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library; the product (kmercamel_amd/) never links, imports or executes it.
//
// A CPU restatement (own code, g++ -std=c++17; integer arithmetic only) of the BIGREEDY hot path of KmerCamel
// v2.3.0.  Every function cites the reference file:line (relative to /root/reference) whose behaviour it restates.
// PINNING: this restatement is checked (tests/test_oracle_vs_reference.py, run where /root/reference exists)
// against oracle/_ref = the reference's own headers compiled in place, and (everywhere) against the golden vectors
// in tests/golden/ that were generated from oracle/_ref by tests/golden/make_golden.py.
//
// Two simplitig orders are provided:
//   mode 0 "reference-order": faithful emulation of the khash bucket layout (khash.h:215-359) so that the seed
//          order of get_simplitigs (simplitigs.h:191-205, khash_utils.h:121-128) -- and therefore the whole
//          output -- is byte-identical to the reference.
//   mode 1 "canonical-order": the layout-free, scheduling-free rule the GPU implements (DESIGN.md section
//          "Canonical simplitigs"): greedy port matching by a symmetric edge priority, chains started at the
//          end with the smaller key, ordered by that key.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <stdexcept>
#include <fstream>
#include <chrono>
#include <zlib.h>

namespace {

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------------------------
// 256-bit unsigned (only what the path needs). Reference: src/uint256_t/uint256_t.h (shifts 114-182, compare, ~).
struct U256 {
    u128 lo, hi;
    U256() : lo(0), hi(0) {}
    U256(u64 v) : lo(v), hi(0) {}
    U256(u128 h, u128 l) : lo(l), hi(h) {}
    U256 operator<<(int s) const {
        if (s == 0) return *this;
        if (s >= 256) return U256();
        if (s >= 128) return U256(lo << (s - 128), 0);
        return U256((hi << s) | (lo >> (128 - s)), lo << s);
    }
    U256 operator>>(int s) const {
        if (s == 0) return *this;
        if (s >= 256) return U256();
        if (s >= 128) return U256(0, hi >> (s - 128));
        return U256(hi >> s, (lo >> s) | (hi << (128 - s)));
    }
    U256 operator|(const U256 &o) const { return U256(hi | o.hi, lo | o.lo); }
    U256 operator&(const U256 &o) const { return U256(hi & o.hi, lo & o.lo); }
    U256 operator^(const U256 &o) const { return U256(hi ^ o.hi, lo ^ o.lo); }
    U256 operator~() const { return U256(~hi, ~lo); }
    U256 operator-(const U256 &o) const {
        U256 r; r.lo = lo - o.lo; r.hi = hi - o.hi - (lo < o.lo ? 1 : 0); return r;
    }
    bool operator==(const U256 &o) const { return lo == o.lo && hi == o.hi; }
    bool operator!=(const U256 &o) const { return !(*this == o); }
    bool operator<(const U256 &o) const { return hi < o.hi || (hi == o.hi && lo < o.lo); }
    bool operator<=(const U256 &o) const { return !(o < *this); }
};

template <typename K> struct KT;
template <> struct KT<u64> {
    static u64 low64(u64 x) { return x; }
    static void to_words(u64 x, u64 *w) { w[0] = x; w[1] = w[2] = w[3] = 0; }
    static u64 from_words(const u64 *w) { return w[0]; }
    static u64 none() { return ~(u64)0; }
};
template <> struct KT<u128> {
    static u64 low64(u128 x) { return (u64)x; }
    static void to_words(u128 x, u64 *w) { w[0] = (u64)x; w[1] = (u64)(x >> 64); w[2] = w[3] = 0; }
    static u128 from_words(const u64 *w) { return ((u128)w[1] << 64) | w[0]; }
    static u128 none() { return ~(u128)0; }
};
template <> struct KT<U256> {
    static u64 low64(const U256 &x) { return (u64)x.lo; }
    static void to_words(const U256 &x, u64 *w) {
        w[0] = (u64)x.lo; w[1] = (u64)(x.lo >> 64); w[2] = (u64)x.hi; w[3] = (u64)(x.hi >> 64);
    }
    static U256 from_words(const u64 *w) { return U256(((u128)w[3] << 64) | w[2], ((u128)w[1] << 64) | w[0]); }
    static U256 none() { return ~U256(); }
};

// ---------------------------------------------------------------------------------------------------------
// A1: nucleotide table (kmers.h:15-32): A,a=0 C,c=1 G,g=2 T,t=3, everything else 4.
inline int nuc(unsigned char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
const char LETTERS[4] = {'A', 'C', 'G', 'T'};  // kmers.h:97

// A5: kmers.h:36-44, 99-102, 124-127
template <typename K> K kmask(int k) { return (K(1) << (2 * k)) - K(1); }
template <typename K> K bit_prefix(K x, int k, int d) { return x >> (2 * (k - d)); }
template <typename K> K bit_suffix(K x, int d) { return x & ((K(1) << (2 * d)) - K(1)); }
template <typename K> int at_index(K x, int k, int i) { return (int)(KT<K>::low64(x >> (2 * (k - i - 1))) & 3); }
inline char masked(char c, bool m) {
    bool upper = c <= 'Z';
    if (m) return upper ? c : (char)(c - ('a' - 'A'));
    return upper ? (char)(c + ('a' - 'A')) : c;
}

// A4: ReverseComplement (kmers.h:93-95): reverse the k 2-bit groups and complement each. Restated base-by-base.
template <typename K> K rev_comp(K x, int k) {
    K r = K(0);
    for (int i = 0; i < k; ++i) {
        r = (r << 2) | K(3 - (KT<K>::low64(x) & 3));
        x = x >> 2;
    }
    return r;
}

// A6 hashes: Wang (khash.h:417-426) on the folded key (khash_utils.h:10-13; note `>>`/`<<` bind before `^`).
inline u64 wang(u64 key) {
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}
inline u64 ref_hash(u64 x) { return wang(x); }
inline u64 ref_hash(u128 x) { return wang((u64)((x >> 65) ^ x ^ (x << 21))); }
inline u64 ref_hash(const U256 &x) {
    U256 f = (x >> 129) ^ x ^ (x << 35);
    return ref_hash((u128)f.lo);
}

// ---------------------------------------------------------------------------------------------------------
// A6: khash set with the reference's bucket layout (khash.h:215-359). state: 0 live, 1 deleted, 2 empty.
template <typename K> struct KhSet {
    u64 n_buckets = 0, size = 0, n_occupied = 0, upper_bound = 0;
    std::vector<uint8_t> st;
    std::vector<K> keys;
    std::vector<uint8_t> vals;   // only used when the table plays the Q-map (k-mer -> saturating count - 1, khash_utils.h:30,34,38)
    static u64 roundup(u64 x) { --x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; x |= x >> 32; return ++x; }
    // kh_get (khash.h:232-246)
    u64 get(const K &key) const {
        if (!n_buckets) return 0;
        u64 mask = n_buckets - 1, i = ref_hash(key) & mask, last = i, step = 0;
        while (st[i] != 2 && (st[i] == 1 || !(keys[i] == key))) {
            i = (i + (++step)) & mask;
            if (i == last) return n_buckets;
        }
        return st[i] != 0 ? n_buckets : i;
    }
    bool contains(const K &key) const { return n_buckets && get(key) != n_buckets; }
    // kh_resize (khash.h:247-310): rehash old buckets 0..n-1 in order with the kick-out chain.
    void resize(u64 new_n) {
        new_n = roundup(new_n);
        if (new_n < 4) new_n = 4;
        if (size >= (u64)(new_n * 0.77 + 0.5)) return;
        std::vector<uint8_t> nst(new_n, 2);
        if (n_buckets < new_n) { keys.resize(new_n); vals.resize(new_n); }
        u64 nmask = new_n - 1;
        for (u64 j = 0; j != n_buckets; ++j) {
            if (st[j] != 0) continue;
            K key = keys[j];
            uint8_t val = vals[j];
            st[j] = 1;
            while (true) {
                u64 i = ref_hash(key) & nmask, step = 0;
                while (nst[i] != 2) i = (i + (++step)) & nmask;
                nst[i] = 0;
                if (i < n_buckets && st[i] == 0) {
                    std::swap(keys[i], key);
                    std::swap(vals[i], val);
                    st[i] = 1;
                } else {
                    keys[i] = key;
                    vals[i] = val;
                    break;
                }
            }
        }
        if (n_buckets > new_n) { keys.resize(new_n); vals.resize(new_n); }
        st.swap(nst);
        n_buckets = new_n;
        n_occupied = size;
        upper_bound = (u64)(n_buckets * 0.77 + 0.5);
    }
    // kh_put (khash.h:311-352). Returns true if newly inserted. *where receives the bucket.
    bool put(const K &key, u64 *where = nullptr) {
        if (n_occupied >= upper_bound) {
            if (n_buckets > (size << 1)) resize(n_buckets - 1);
            else resize(n_buckets + 1);
        }
        u64 mask = n_buckets - 1, site = n_buckets, x = n_buckets, i = ref_hash(key) & mask, step = 0;
        if (st[i] == 2) x = i;
        else {
            u64 last = i;
            while (st[i] != 2 && (st[i] == 1 || !(keys[i] == key))) {
                if (st[i] == 1) site = i;
                i = (i + (++step)) & mask;
                if (i == last) { x = site; break; }
            }
            if (x == n_buckets) {
                if (st[i] == 2 && site != n_buckets) x = site;
                else x = i;
            }
        }
        if (where) *where = x;
        if (st[x] == 2) { keys[x] = key; st[x] = 0; ++size; ++n_occupied; return true; }
        if (st[x] == 1) { keys[x] = key; st[x] = 0; ++size; return true; }
        return false;
    }
    // kh_del (khash.h:353-359)
    void del(u64 x) { if (x != n_buckets && st[x] == 0) { st[x] = 1; --size; } }
};

// ---------------------------------------------------------------------------------------------------------
// A3: record reader with kseq semantics (kseq.h:183-224), over a fully decompressed buffer (zlib handles
// plain and gz transparently like gzdopen/gzread in parser.h:88-101).
bool slurp(const std::string &path, std::string &buf) {
    gzFile f = (path == "-") ? gzdopen(0, "r") : gzopen(path.c_str(), "r");
    if (!f) return false;
    char tmp[1 << 16];
    int n;
    while ((n = gzread(f, tmp, sizeof tmp)) > 0) buf.append(tmp, n);
    gzclose(f);
    return true;
}

struct Record { std::string name, comment, seq; };

std::vector<Record> parse_records(const std::string &b) {
    std::vector<Record> out;
    size_t p = 0, n = b.size();
    int last_char = 0;
    auto take_line = [&](std::string &dst) {  // ks_getuntil2(KS_SEP_LINE, append): strips one trailing '\r' if dst.size()>1
        size_t e = b.find('\n', p);
        if (e == std::string::npos) e = n;
        dst.append(b, p, e - p);
        if (dst.size() > 1 && dst.back() == '\r') dst.pop_back();
        p = e < n ? e + 1 : n;
    };
    while (true) {
        if (last_char == 0) {  // jump to the next header char (kseq.h:188-192)
            while (p < n && b[p] != '>' && b[p] != '@') ++p;
            if (p >= n) break;
            last_char = b[p++];
        }
        if (p >= n) break;  // reading the name hits EOF with nothing read: -1
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
        while (p < n) {  // kseq.h:199-203
            c = (unsigned char)b[p++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c == '\n') { c = -1; continue; }
            r.seq.push_back((char)c);
            take_line(r.seq);
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;
        if (c != '+') {  // FASTA record
            out.push_back(r);
            if (c == -1) break;  // EOF: the next kseq_read returns -1
            continue;
        }
        // FASTQ (kseq.h:212-223)
        size_t e = b.find('\n', p);
        if (e == std::string::npos) break;  // -2: no quality string; ReadKMers' loop stops (parser.h:112)
        p = e + 1;
        std::string qual;
        while (p < n) {
            take_line(qual);
            if (!(qual.size() < r.seq.size())) break;
        }
        last_char = 0;
        if (qual.size() != r.seq.size()) break;  // -2
        out.push_back(r);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------
// A2: AddKMers (parser.h:22-49).
template <typename K, typename F>
void for_each_kmer(const char *seq, size_t len, int k, bool complements, bool case_sensitive, F &&emit) {
    int64_t cur_len = 0;
    K cur = K(0), rc = K(0), cases = K(0);
    K mask = kmask<K>(k);
    int shift = 2 * (k - 1);
    for (size_t i = 0; i < len; ++i) {
        int d = nuc((unsigned char)seq[i]);
        if (d >= 4) { cur = rc = K(0); cur_len = 0; continue; }
        cur = ((cur << 2) | K((u64)d)) & mask;
        rc = (rc >> 2) | (K((u64)(3 ^ d)) << shift);
        cases = (cases | K((u64)(!case_sensitive || seq[i] <= 'Z'))) << 1;
        if (++cur_len >= k && !((cases & (K(1) << k)) == K(0))) {
            K canonical = (!complements || cur < rc) ? cur : rc;
            emit(canonical, i);
        }
    }
}

template <typename K>
void add_kmers(KhSet<K> &set, const char *seq, size_t len, int k, bool complements, bool case_sensitive) {
    for_each_kmer<K>(seq, len, k, complements, case_sensitive, [&](const K &x, size_t) { set.put(x); });
}

// AddKMersWithFrequencies (parser.h:51-85): value = min(255, occurrences) - 1
template <typename K>
void add_kmers_with_frequencies(KhSet<K> &map, const char *seq, size_t len, int k, bool complements) {
    for_each_kmer<K>(seq, len, k, complements, false, [&](const K &x, size_t) {
        u64 it = map.get(x);
        if (map.n_buckets && it != map.n_buckets) {
            unsigned count = map.vals[it];
            map.vals[it] = (uint8_t)std::min(255u, count + 1);
        } else {
            u64 w = 0;
            map.put(x, &w);
            map.vals[w] = 0;
        }
    });
}
// kMersToVecFiltered + kMersFromVec (khash_utils.h:143-159, parser.h:122-141): survivors in bucket order into a fresh set
template <typename K> void filter_into_set(const KhSet<K> &map, unsigned min_frequency, KhSet<K> &set) {
    for (u64 i = 0; i < map.n_buckets; ++i)
        if (map.st[i] == 0 && (unsigned)map.vals[i] + 1 >= min_frequency) set.put(map.keys[i]);
}

typedef std::vector<uint8_t> Simplitig;  // base codes 0..3 (B1: simplitigs.h:11 stores the same as 2 bits/nt MSB first)

template <typename K> K kmer_at(const Simplitig &s, int k, size_t index) {  // simplitigs.h:31-39
    K r = K(0);
    for (int i = 0; i < k; ++i) r = (r << 2) | K((u64)s[index + i]);
    return r;
}
Simplitig simplitig_rc(const Simplitig &s) {  // simplitigs.h:14-21
    Simplitig r(s.size());
    for (size_t i = 0; i < s.size(); ++i) r[s.size() - 1 - i] = 3 - s[i];
    return r;
}
std::string simplitig_str(const Simplitig &s) {
    std::string r(s.size(), 'N');
    for (size_t i = 0; i < s.size(); ++i) r[i] = LETTERS[s[i]];
    return r;
}
bool simplitig_from_str(const char *str, Simplitig &s) {  // simplitigs.h:78-87 (assert ACGT)
    size_t n = strlen(str);
    s.resize(n);
    for (size_t i = 0; i < n; ++i) { int d = nuc((unsigned char)str[i]); if (d > 3) return false; s[i] = (uint8_t)d; }
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// B2-B4, reference order: simplitig_right_rev_extension / next_simplitig / get_simplitigs (simplitigs.h:105-205),
// eraseKMer / nextKMer (khash_utils.h:106-128).
template <typename K>
int right_rev_extension(bool complements, K &forward, K &backward, KhSet<K> &set, int k) {
    bool forward_direction = complements ? true : (backward == KT<K>::none());
    K mask = kmask<K>(k);
    K fbase = (forward << 2) & mask, bbase = backward >> 2;
    for (int ext = 0; ext < 4; ++ext) {
        K nf = fbase | K((u64)(3 ^ ext));
        K nb = bbase | (K((u64)ext) << (2 * (k - 1)));
        K canonical = complements ? ((nf <= nb) ? nf : nb) : (forward_direction ? nf : nb);
        u64 it = set.get(canonical);
        if (set.n_buckets && it != set.n_buckets) {
            if (complements) { backward = nb; forward = nf; }
            else if (forward_direction) forward = nf;
            else backward = nb;
            set.del(it);
            return ext;
        }
    }
    return -1;
}

template <typename K>
Simplitig next_simplitig(bool complements, KhSet<K> &set, K begin, int k) {
    K last = begin, first = begin;
    K last_c = rev_comp(begin, k), first_c = last_c;
    if (!complements) first_c = last_c = KT<K>::none();
    Simplitig back(k), front;
    for (int i = 0; i < k; ++i) back[i] = (uint8_t)at_index(begin, k, i);
    {   // eraseKMer (khash_utils.h:106-117)
        u64 it = set.get(last);
        if (it != set.n_buckets) set.del(it);
        if (complements) { it = set.get(rev_comp(last, k)); if (it != set.n_buckets) set.del(it); }
    }
    while (true) {
        int ext = right_rev_extension(complements, last, last_c, set, k);
        if (ext < 0) break;
        back.push_back((uint8_t)(3 ^ ext));  // pushes !(ext&2), !(ext&1)
    }
    while (true) {
        int ext = right_rev_extension(complements, first_c, first, set, k);
        if (ext < 0) break;
        front.push_back((uint8_t)ext);  // pushes (ext&1, ext&2) then the whole bit vector is reversed
    }
    std::reverse(front.begin(), front.end());
    front.insert(front.end(), back.begin(), back.end());
    return front;
}

template <typename K>
std::vector<Simplitig> simplitigs_reference_order(KhSet<K> &set, int k, bool complements) {
    std::vector<Simplitig> res;
    u64 last_index = 0;
    while (true) {
        while (last_index < set.n_buckets && set.st[last_index] != 0) ++last_index;
        if (last_index >= set.n_buckets) return res;
        res.push_back(next_simplitig(complements, set, set.keys[last_index], k));
    }
}

// ---------------------------------------------------------------------------------------------------------
// Canonical-order simplitigs (the GPU's rule; DESIGN.md "Canonical simplitigs"). Layout-free.
inline u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
template <typename K> u64 key_hash(const K &x) {
    u64 w[4];
    KT<K>::to_words(x, w);
    const int W = sizeof(K) / 8;
    u64 h = mix64(w[W - 1]);
    for (int i = W - 2; i >= 0; --i) h = mix64(h ^ w[i]);
    return h;
}
inline u64 port_term(u64 keyhash, int port) { return mix64(keyhash ^ (port ? 0x9e3779b97f4a7c15ULL : 0x7f4a7c159e3779b9ULL)); }

template <typename K> struct Edge {
    u64 pr;
    K ka, kb;      // (ka,pa) <= (kb,pb)
    uint8_t pa, pb;
    uint32_t ia, ib;
};

template <typename K> struct KeyHasher { size_t operator()(const K &x) const { return (size_t)key_hash(x); } };

template <typename K>
std::vector<Simplitig> simplitigs_canonical_order(std::vector<K> keys, int k, bool complements) {
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    size_t n = keys.size();
    std::unordered_map<K, uint32_t, KeyHasher<K>> idx;
    idx.reserve(n * 2);
    for (size_t i = 0; i < n; ++i) idx[keys[i]] = (uint32_t)i;
    K mask = kmask<K>(k);
    std::vector<Edge<K>> edges;
    auto add_edge = [&](uint32_t ix, int px, const K &y, int py) {
        auto it = idx.find(y);
        if (it == idx.end()) return;
        uint32_t iy = it->second;
        const K &x = keys[ix];
        bool x_first = (x < y) || (x == y && px <= py);
        Edge<K> e;
        if (x_first) { e.ka = x; e.pa = (uint8_t)px; e.ia = ix; e.kb = y; e.pb = (uint8_t)py; e.ib = iy; }
        else { e.ka = y; e.pa = (uint8_t)py; e.ia = iy; e.kb = x; e.pb = (uint8_t)px; e.ib = ix; }
        e.pr = mix64(port_term(key_hash(e.ka), e.pa) + port_term(key_hash(e.kb), e.pb));
        edges.push_back(e);
    };
    for (uint32_t i = 0; i < n; ++i) {
        const K x = keys[i];
        if (complements) {
            bool pal = (x == rev_comp(x, k));
            for (int c = 0; c < 4; ++c) {
                if (!pal) {  // port R (1): append c
                    K nf = ((x << 2) & mask) | K((u64)c), r = rev_comp(nf, k);
                    K y = nf < r ? nf : r;
                    if (!(y == x)) add_edge(i, 1, y, (nf == r) ? 0 : ((y == nf) ? 0 : 1));
                }
                {   // port L (0): prepend c
                    K nb = (x >> 2) | (K((u64)c) << (2 * (k - 1))), r = rev_comp(nb, k);
                    K y = nb < r ? nb : r;
                    if (!(y == x)) add_edge(i, 0, y, (nb == r) ? 0 : ((y == nb) ? 1 : 0));
                }
            }
        } else {
            for (int c = 0; c < 4; ++c) {
                K y = ((x << 2) & mask) | K((u64)c);
                if (!(y == x)) add_edge(i, 1, y, 0);
            }
        }
    }
    auto edge_less = [](const Edge<K> &a, const Edge<K> &b) {  // "a is preferred over b"
        if (a.pr != b.pr) return a.pr > b.pr;
        if (!(a.ka == b.ka)) return a.ka < b.ka;
        if (a.pa != b.pa) return a.pa < b.pa;
        if (!(a.kb == b.kb)) return a.kb < b.kb;
        return a.pb < b.pb;
    };
    std::sort(edges.begin(), edges.end(), edge_less);
    // (in bi mode every edge was generated from both endpoints; duplicates are adjacent and harmless to greedy)
    const uint32_t NONE = 0xffffffffu;
    std::vector<uint32_t> link(2 * n, NONE);      // link[2*i+port] = 2*j+port' of the matched port
    for (auto &e : edges) {
        uint32_t a = 2 * e.ia + e.pa, b = 2 * e.ib + e.pb;
        if (a == b) continue;
        if (link[a] == NONE && link[b] == NONE) { link[a] = b; link[b] = a; }
    }
    // cycles (see below)
    std::vector<uint8_t> dead_before(2 * n, 0);   // ports dead after the matching (ports opened by a cut do not count: the rule is independent of the order of the cycles)
    for (size_t pidx = 0; pidx < 2 * n; ++pidx) dead_before[pidx] = link[pidx] == NONE;
    std::vector<uint8_t> seen(n, 0);
    size_t cycles_cut = 0;
    auto step = [&](uint32_t node, int o, uint32_t &nnode, int &no) -> bool {  // o=0: enter L leave R; o=1: enter R leave L
        uint32_t out = link[2 * node + (o == 0 ? 1 : 0)];
        if (out == NONE) return false;
        nnode = out >> 1;
        no = (out & 1) == 0 ? 0 : 1;  // entering on port L => forward
        return true;
    };
    for (uint32_t i = 0; i < n; ++i) {   // mark everything reachable from a free end
        for (int o = 0; o < 2; ++o) {
            if (link[2 * i + (o == 0 ? 0 : 1)] != NONE) continue;
            uint32_t cur = i; int co = o;
            while (true) { seen[cur] = 1; uint32_t nx; int no; if (!step(cur, co, nx, no)) break; cur = nx; co = no; }
        }
    }
    // Where a cycle is opened: the reference's delete-as-you-go walk never closes a cycle (simplitigs.h:161-183) -- it would have
    // run on from one of these k-mers into a neighbouring chain.  So the candidate edges from a port x.p on a cycle to a port y.q
    // that the matching left DEAD are taken in the matching's own order of preference (the order of `edges`): if x's cycle is
    // still closed and y.q still free, the cycle is opened at x.p and x.p is matched with y.q -- the cycle becomes the
    // continuation of that chain, as in the reference's walk.  (Ports that die by opening a cycle do not count: the rule depends
    // on the matching only, not on an order of the cycles.)  A cycle no such edge opens is cut at its minimum key on port L.
    std::vector<uint32_t> cyc_of(n, NONE), cyc_min;
    for (uint32_t i = 0; i < n; ++i) {
        if (seen[i]) continue;
        const uint32_t id = (uint32_t)cyc_min.size();
        uint32_t cur = i, best = i; int co = 0;
        while (true) {
            seen[cur] = 1; cyc_of[cur] = id;
            if (keys[cur] < keys[best]) best = cur;
            uint32_t nx; int no; step(cur, co, nx, no);
            cur = nx; co = no;
            if (cur == i) break;
        }
        cyc_min.push_back(best);
    }
    std::vector<uint8_t> opened(cyc_min.size(), 0);
    if (!cyc_min.empty()) {
        for (auto &e : edges) {
            const uint32_t a = 2 * e.ia + e.pa, b = 2 * e.ib + e.pb;
            if (a == b) continue;
            for (int sw = 0; sw < 2; ++sw) {
                const uint32_t x = sw ? b : a, y = sw ? a : b;
                const uint32_t id = cyc_of[x >> 1];
                if (id == NONE || opened[id] || !dead_before[y] || link[y] != NONE) continue;
                const uint32_t other = link[x];
                link[other] = NONE;
                link[x] = y; link[y] = x;
                opened[id] = 1;
                ++cycles_cut;
                break;
            }
        }
    }
    for (size_t id = 0; id < cyc_min.size(); ++id) {
        if (opened[id]) continue;
        const uint32_t cutport = 2 * cyc_min[id] + 0, other = link[cutport];
        link[cutport] = NONE;
        link[other] = NONE;
        ++cycles_cut;
    }
    if (getenv("KC_ORACLE_DEBUG")) fprintf(stderr, "[oracle] cycles cut %zu\n", cycles_cut);
    // chains: start at the end with the smaller key; order by that key.
    struct Chain { K head; uint32_t node; int o; };
    std::vector<Chain> chains;
    for (uint32_t i = 0; i < n; ++i) {
        for (int o = 0; o < 2; ++o) {
            if (link[2 * i + (o == 0 ? 0 : 1)] != NONE) continue;
            if (!complements && o == 1) continue;
            uint32_t cur = i; int co = o; size_t len = 1;
            while (true) { uint32_t nx; int no; if (!step(cur, co, nx, no)) break; cur = nx; co = no; ++len; }
            bool winner = complements ? (len == 1 ? (o == 0) : (keys[i] < keys[cur])) : true;
            if (winner) chains.push_back({keys[i], i, o});
        }
    }
    std::sort(chains.begin(), chains.end(), [](const Chain &a, const Chain &b) { return a.head < b.head; });
    std::vector<Simplitig> res;
    res.reserve(chains.size());
    for (auto &ch : chains) {
        Simplitig s;
        uint32_t cur = ch.node; int co = ch.o;
        K first = (co == 0) ? keys[cur] : rev_comp(keys[cur], k);
        for (int i = 0; i < k; ++i) s.push_back((uint8_t)at_index(first, k, i));
        while (true) {
            uint32_t nx; int no;
            if (!step(cur, co, nx, no)) break;
            cur = nx; co = no;
            const K &y = keys[cur];
            s.push_back(co == 0 ? (uint8_t)(KT<K>::low64(y) & 3) : (uint8_t)(3 - at_index(y, k, 0)));
        }
        res.push_back(std::move(s));
    }
    return res;
}

// ---------------------------------------------------------------------------------------------------------
// C1: OverlapHamiltonianPath (global.h:43-133). End k-mers are extracted once (the reference re-extracts them
// through accessPrefix/accessSuffix on every access, global.h:16-21 -- same values).
struct OverlapPath { std::vector<u64> edge_from; std::vector<uint8_t> overlaps; };
const u64 NONE64 = ~(u64)0;

template <typename K>
OverlapPath overlap_path_from_ends(const std::vector<K> &firstK, const std::vector<K> &lastK, int k, bool complements,
                                   bool lower_bound) {
    size_t n = firstK.size();
    size_t V = n * (1 + (complements ? 1 : 0));
    size_t batch = V / 16 + 1;  // MEMORY_REDUCTION_FACTOR = 16 (global.h:33,49)
    std::vector<K> pre(V), suf(V);
    for (size_t i = 0; i < n; ++i) {
        pre[i] = firstK[i]; suf[i] = lastK[i];
        if (complements) { pre[i + n] = rev_comp(lastK[i], k); suf[i + n] = rev_comp(firstK[i], k); }
    }
    OverlapPath res;
    res.edge_from.assign(V, NONE64);
    res.overlaps.assign(V, 0xff);
    std::vector<uint8_t> sufForbidden(V, 0), preForbidden(V, 0);
    std::vector<u64> first(n), last(n), next(batch);
    for (size_t i = 0; i < n; ++i) first[i] = last[i] = i;
    auto acc = [&](const std::vector<u64> &a, const std::vector<u64> &b, u64 index) -> u64 {  // accessFirstLast (global.h:28-29)
        return n > index ? a[index] : (b[index - n] + n) % (2 * n);
    };
    std::unordered_map<K, u64, KeyHasher<K>> prefixes;
    prefixes.reserve(batch * 2);
    for (int d = k - 1; d >= 0; --d) {
        for (int part = 0; part < 16; ++part) {
            prefixes.clear();
            size_t from = part * batch, to = std::min(V, (size_t)(part + 1) * batch);
            for (size_t i = from; i < to; ++i) {
                if (preForbidden[i]) continue;
                K key = bit_prefix(pre[i], k, d);
                auto it = prefixes.find(key);
                if (it != prefixes.end()) { next[i - from] = it->second; it->second = i; }
                else { next[i - from] = NONE64; prefixes.emplace(key, i); }
            }
            if (prefixes.empty()) continue;
            for (size_t i = 0; i < V; ++i) {
                if (sufForbidden[i]) continue;
                K key = bit_suffix(suf[i], d);
                auto it = prefixes.find(key);
                if (it == prefixes.end()) continue;
                u64 previous, j;
                previous = j = it->second;
                while (j != NONE64 && ((!lower_bound && (i + n) % (2 * n) == j) ||
                                       (!lower_bound && acc(first, last, i) == j) || preForbidden[j])) {
                    u64 new_j = next[j - from];
                    if (preForbidden[j]) next[previous - from] = new_j;
                    else previous = j;
                    j = new_j;
                }
                if (j == NONE64) continue;
                u64 ex[2] = {i, (j + n) % V}, ey[2] = {j, (i + n) % V};
                for (int e = 0; e < (complements ? 2 : 1); ++e) {
                    u64 x = ex[e], y = ey[e];
                    res.edge_from[x] = y;
                    res.overlaps[x] = (uint8_t)d;
                    preForbidden[y] = 1;
                    u64 lastY = acc(last, first, y), firstX = acc(first, last, x);
                    if (lastY < n) first[lastY] = firstX;
                    if (firstX < n) last[firstX] = lastY;
                    sufForbidden[x] = 1;
                }
                next[previous - from] = next[j - from];
            }
        }
    }
    return res;
}

template <typename K>
OverlapPath overlap_path(const std::vector<Simplitig> &s, int k, bool complements, bool lower_bound) {
    std::vector<K> f(s.size()), l(s.size());
    for (size_t i = 0; i < s.size(); ++i) { f[i] = kmer_at<K>(s[i], k, 0); l[i] = kmer_at<K>(s[i], k, s[i].size() - k); }
    return overlap_path_from_ends(f, l, k, complements, lower_bound);
}

// C2: SuperstringFromPath (global.h:136-210), fill_kmers (simplitigs.h:41-51), containsKMer (khash_utils.h:98-103).
template <typename K>
u64 superstring_from_path(const OverlapPath &hp, const std::vector<Simplitig> &s, int k, bool complements, std::string &out,
                          std::string *mask_out) {
    size_t n = s.size(), V = n * (1 + (complements ? 1 : 0));
    std::vector<uint8_t> is_start(V, 1);
    for (u64 e : hp.edge_from) if (e != NONE64) is_start[e] = 0;
    size_t start = 0;
    while (start < V && !is_start[start]) ++start;
    std::unordered_map<K, char, KeyHasher<K>> dict;
    if (mask_out) for (auto &x : s) for (size_t i = 0; i + k <= x.size(); ++i) dict.emplace(kmer_at<K>(x, k, i), 1);
    auto contains = [&](const K &x) {
        if (dict.count(x)) return true;
        return complements && dict.count(rev_comp(x, k)) > 0;
    };
    auto vertex = [&](size_t v) { return v < n ? s[v] : simplitig_rc(s[v - n]); };
    Simplitig cur = vertex(start);
    auto print_start = [&](const Simplitig &x) {
        for (size_t i = 0; i + k <= x.size(); ++i) { out.push_back(LETTERS[x[i]]); if (mask_out) mask_out->push_back(LETTERS[x[i]]); }
    };
    K last = kmer_at<K>(cur, k, cur.size() - k);
    print_start(cur);
    u64 length = cur.size();
    while (hp.edge_from[start] != NONE64) {
        int ov = hp.overlaps[start];
        cur = vertex(hp.edge_from[start]);
        length += cur.size() - ov;
        for (int j = 1; j <= k - 1 - ov; ++j) out.push_back((char)tolower(LETTERS[at_index(last, k, j)]));
        if (mask_out) {
            K first_next = kmer_at<K>(cur, k, 0);
            for (int j = 0; j < k - ov - 1; ++j) {
                last = ((last << 2) | K((u64)at_index(first_next, k, ov + j))) & kmask<K>(k);
                mask_out->push_back(masked(LETTERS[at_index(last, k, 0)], contains(last)));
            }
        }
        last = kmer_at<K>(cur, k, cur.size() - k);
        print_start(cur);
        start = hp.edge_from[start];
    }
    std::string tail;
    for (int j = 1; j <= k - 1; ++j) tail.push_back((char)tolower(LETTERS[at_index(last, k, j)]));
    out += tail;
    if (mask_out) { *mask_out += tail; mask_out->push_back('\n'); }
    return length;
}

// D1: PartialPreSort (global_sparse.h:14-35) + sparse overlap path / superstring (43-202) via 1-k-mer simplitigs:
// OverlapHamiltonianPathSparse is C1 over single k-mers (prefix = suffix = the k-mer) and SuperstringFromPathSparse
// prints exactly what C2 prints for simplitigs of length k.
template <typename K> void partial_presort(std::vector<K> &v, int k) {
    int bits = std::min(2 * k, 8);
    int shift = 2 * k - bits;
    std::stable_sort(v.begin(), v.end(), [&](const K &a, const K &b) {
        return (KT<K>::low64(a >> shift) & ((1u << bits) - 1)) < (KT<K>::low64(b >> shift) & ((1u << bits) - 1));
    });
}

// next-row 1: OptimizeOnes (masks.h:40-78) after AddKMers(case_sensitive=true) (masks.h:243).
template <typename K>
int maskopt_ones(const std::string &ms, int k, bool complements, bool minimize, std::string &out) {
    KhSet<K> set;
    add_kmers(set, ms.data(), ms.size(), k, complements, true);
    K cur = K(0), rc = K(0), mask = kmask<K>(k);
    int shift = 2 * (k - 1), validation = 0;
    for (size_t i = 0; i < ms.size(); ++i) {
        int d = nuc((unsigned char)ms[i]);
        validation |= d;
        cur = ((cur << 2) | K((u64)d)) & mask;
        rc = (rc >> 2) | (K((u64)(3 ^ d)) << shift);
        if (i + 1 >= (size_t)k) {
            K canonical = (!complements || cur < rc) ? cur : rc;
            u64 it = set.get(canonical);
            bool contained = set.n_buckets && it != set.n_buckets;
            out.push_back(masked(ms[i - k + 1], contained));
            if (minimize && contained) set.del(it);
        }
    }
    for (size_t i = ms.size() - k + 1; i < ms.size(); ++i) out.push_back(masked(ms[i], false));
    return validation >= 4 ? -3 : 0;
}

// ---------------------------------------------------------------------------------------------------------
template <typename F> auto dispatch(int k, F &&f) {
    if (k < 32) return f(u64(0));          // main.cpp:309-315
    if (k < 64) return f(u128(0));
    return f(U256());
}

char *dup_string(const std::string &s) {
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}
std::string join(const std::vector<Simplitig> &v) {
    std::string r;
    for (auto &x : v) { r += simplitig_str(x); r += '\n'; }
    return r;
}
std::vector<Simplitig> from_strings(const char **strs, u64 n) {
    std::vector<Simplitig> v(n);
    for (u64 i = 0; i < n; ++i) simplitig_from_str(strs[i], v[i]);
    return v;
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename K>
std::vector<Simplitig> build_simplitigs(KhSet<K> &set, int k, bool complements, int mode) {
    if (mode == 0) return simplitigs_reference_order(set, k, complements);
    std::vector<K> keys;
    keys.reserve(set.size);
    for (u64 i = 0; i < set.n_buckets; ++i) if (set.st[i] == 0) keys.push_back(set.keys[i]);
    return simplitigs_canonical_order(keys, k, complements);
}

// Greedy branch of kmercamel<>() (main.cpp:122-212) without the FASTA header lines.
template <typename K>
int compute(const std::string &path, int k, bool complements, bool assume_simplitigs, int mode, std::string &out,
            std::string *mask_out, u64 *stats, unsigned min_frequency = 1) {
    std::string buf;
    if (!slurp(path, buf)) return -1;  // parser.h:96 throws invalid_argument("couldn't open file ...")
    double t0 = now_s();
    auto recs = parse_records(buf);
    std::vector<Simplitig> simplitigs;
    u64 kmer_count = 0;
    double t1 = t0;
    if (!assume_simplitigs) {
        KhSet<K> set;
        if (min_frequency <= 1) {
            for (auto &r : recs) add_kmers(set, r.seq.data(), r.seq.size(), k, complements, false);
        } else {   // ReadKMersFiltered (parser.h:122-141)
            KhSet<K> freq;
            for (auto &r : recs) add_kmers_with_frequencies(freq, r.seq.data(), r.seq.size(), k, complements);
            filter_into_set(freq, min_frequency, set);
        }
        if (!set.size) return 1;
        kmer_count = set.size;
        t1 = now_s();
        simplitigs = build_simplitigs(set, k, complements, mode);
    } else {
        for (auto &r : recs) {
            Simplitig s;
            if (!simplitig_from_str(r.seq.c_str(), s)) return -4;  // assert (simplitigs.h:82)
            simplitigs.push_back(s);
        }
        for (auto &s : simplitigs) kmer_count += s.size() - k + 1;
    }
    double t2 = now_s();
    u64 n_simplitigs = simplitigs.size();
    bool sparse = !assume_simplitigs && simplitigs.size() * 5 >= kmer_count;  // main.cpp:94,175
    if (sparse) {
        std::vector<K> v;
        for (auto &s : simplitigs) for (size_t i = 0; i + k <= s.size(); ++i) v.push_back(kmer_at<K>(s, k, i));
        partial_presort(v, k);
        simplitigs.clear();
        for (auto &x : v) { Simplitig s(k); for (int i = 0; i < k; ++i) s[i] = (uint8_t)at_index(x, k, i); simplitigs.push_back(s); }
    }
    if (simplitigs.empty()) return -2;  // global.h:220 "input cannot be empty"
    auto hp = overlap_path<K>(simplitigs, k, complements, false);
    double t3 = now_s();
    u64 L = superstring_from_path<K>(hp, simplitigs, k, complements, out, mask_out);
    double t4 = now_s();
    if (stats) {
        stats[0] = kmer_count; stats[1] = n_simplitigs; stats[2] = L; stats[3] = sparse;
        stats[4] = (u64)((t1 - t0) * 1e6); stats[5] = (u64)((t2 - t1) * 1e6); stats[6] = (u64)((t3 - t2) * 1e6);
        stats[7] = (u64)((t4 - t3) * 1e6);
    }
    return 0;
}

}  // namespace

extern "C" {

void kco_free(void *p) { free(p); }

// mode: 0 reference-order, 1 canonical-order. stats[8] as in oracle/ref_shim.cpp. *out/*mask_out are the sequence
// lines without header and without trailing newline (mask_out may be NULL).
int kco_compute_file(const char *path, int k, int complements, int assume_simplitigs, int mode, char **out, char **mask_out,
                     uint64_t *stats) {
    std::string o, m;
    uint64_t st[8] = {0};
    int rc = dispatch(k, [&](auto kt) {
        using K = decltype(kt);
        // number of simplitigs before a possible sparse expansion is reported in stats[1]
        return compute<K>(path, k, complements != 0, assume_simplitigs != 0, mode, o, mask_out ? &m : nullptr, st);
    });
    if (rc) return rc;
    if (!m.empty() && m.back() == '\n') m.pop_back();
    *out = dup_string(o);
    if (mask_out) *mask_out = dup_string(m);
    if (stats) memcpy(stats, st, sizeof st);
    return 0;
}

// compute with the -z filter (main.cpp:150-153)
int kco_compute_file_z(const char *path, int k, int complements, int mode, int min_frequency, char **out, char **mask_out, uint64_t *stats) {
    std::string o, m;
    uint64_t st[8] = {0};
    int rc = dispatch(k, [&](auto kt) {
        using K = decltype(kt);
        return compute<K>(path, k, complements != 0, false, mode, o, mask_out ? &m : nullptr, st, (unsigned)min_frequency);
    });
    if (rc) return rc;
    if (!m.empty() && m.back() == '\n') m.pop_back();
    *out = dup_string(o);
    if (mask_out) *mask_out = dup_string(m);
    if (stats) memcpy(stats, st, sizeof st);
    return 0;
}

// ReadKMersFiltered (parser.h:122-141): sorted surviving k-mers of a file
int64_t kco_file_kmer_set_z(const char *path, int k, int complements, int min_frequency, uint64_t **out) {
    std::string buf;
    if (!slurp(path, buf)) return -1;
    auto recs = parse_records(buf);
    return dispatch(k, [&](auto kt) -> int64_t {
        using K = decltype(kt);
        KhSet<K> freq, set;
        for (auto &r : recs) add_kmers_with_frequencies(freq, r.seq.data(), r.seq.size(), k, complements != 0);
        filter_into_set(freq, (unsigned)min_frequency, set);
        std::vector<K> v;
        for (u64 i = 0; i < set.n_buckets; ++i) if (set.st[i] == 0) v.push_back(set.keys[i]);
        std::sort(v.begin(), v.end());
        if (out) {
            u64 *w = (u64 *)malloc(std::max<size_t>(1, v.size()) * 32);
            for (size_t i = 0; i < v.size(); ++i) KT<K>::to_words(v[i], w + 4 * i);
            *out = w;
        }
        return (int64_t)v.size();
    });
}

int64_t kco_kmer_set(const char *seq, uint64_t len, int k, int complements, int case_sensitive, uint64_t **out) {
    return dispatch(k, [&](auto kt) -> int64_t {
        using K = decltype(kt);
        KhSet<K> set;
        add_kmers(set, seq, len, k, complements != 0, case_sensitive != 0);
        std::vector<K> v;
        for (u64 i = 0; i < set.n_buckets; ++i) if (set.st[i] == 0) v.push_back(set.keys[i]);
        std::sort(v.begin(), v.end());
        u64 *w = (u64 *)malloc(std::max<size_t>(1, v.size()) * 32);
        for (size_t i = 0; i < v.size(); ++i) KT<K>::to_words(v[i], w + 4 * i);
        *out = w;
        return (int64_t)v.size();
    });
}

// Sorted distinct canonical k-mers of a whole FASTA/FASTQ(.gz) file (ReadKMers, parser.h:106-118).
int64_t kco_file_kmer_set(const char *path, int k, int complements, int case_sensitive, uint64_t **out) {
    std::string buf;
    if (!slurp(path, buf)) return -1;
    auto recs = parse_records(buf);
    return dispatch(k, [&](auto kt) -> int64_t {
        using K = decltype(kt);
        KhSet<K> set;
        for (auto &r : recs) add_kmers(set, r.seq.data(), r.seq.size(), k, complements != 0, case_sensitive != 0);
        std::vector<K> v;
        for (u64 i = 0; i < set.n_buckets; ++i) if (set.st[i] == 0) v.push_back(set.keys[i]);
        std::sort(v.begin(), v.end());
        if (out) {
            u64 *w = (u64 *)malloc(std::max<size_t>(1, v.size()) * 32);
            for (size_t i = 0; i < v.size(); ++i) KT<K>::to_words(v[i], w + 4 * i);
            *out = w;
        }
        return (int64_t)v.size();
    });
}

int64_t kco_read_kmers_count(const char *path, int k, int complements, int case_sensitive) {
    return kco_file_kmer_set(path, k, complements, case_sensitive, nullptr);
}

// Records of a file as "name\tcomment\tseq\n" lines (kseq semantics).
char *kco_parse_records(const char *path) {
    std::string buf;
    if (!slurp(path, buf)) return nullptr;
    std::string r;
    for (auto &x : parse_records(buf)) r += x.name + "\t" + x.comment + "\t" + x.seq + "\n";
    return dup_string(r);
}

char *kco_simplitigs_of_sequence(const char *seq, uint64_t len, int k, int complements, int mode) {
    return dup_string(dispatch(k, [&](auto kt) -> std::string {
        using K = decltype(kt);
        KhSet<K> set;
        add_kmers(set, seq, len, k, complements != 0, false);
        return join(build_simplitigs(set, k, complements != 0, mode));
    }));
}

char *kco_simplitigs_of_kmers(const uint64_t *words, uint64_t n, int k, int complements, int mode) {
    return dup_string(dispatch(k, [&](auto kt) -> std::string {
        using K = decltype(kt);
        KhSet<K> set;
        for (u64 i = 0; i < n; ++i) set.put(KT<K>::from_words(words + 4 * i));
        return join(build_simplitigs(set, k, complements != 0, mode));
    }));
}

int kco_overlap_path(const char **simplitigs, uint64_t n, int k, int complements, int lower_bound, uint64_t *edge_from,
                     uint8_t *overlaps) {
    return dispatch(k, [&](auto kt) -> int {
        using K = decltype(kt);
        auto hp = overlap_path<K>(from_strings(simplitigs, n), k, complements != 0, lower_bound != 0);
        memcpy(edge_from, hp.edge_from.data(), hp.edge_from.size() * 8);
        memcpy(overlaps, hp.overlaps.data(), hp.overlaps.size());
        return 0;
    });
}

int64_t kco_superstring_from_path(const char **simplitigs, uint64_t n, const uint64_t *edge_from, const uint8_t *overlaps, int k,
                                  int complements, char **out, char **mask_out) {
    return dispatch(k, [&](auto kt) -> int64_t {
        using K = decltype(kt);
        auto v = from_strings(simplitigs, n);
        size_t V = n * (1 + (complements != 0));
        OverlapPath hp;
        hp.edge_from.assign(edge_from, edge_from + V);
        hp.overlaps.assign(overlaps, overlaps + V);
        std::string o, m;
        u64 L = superstring_from_path<K>(hp, v, k, complements != 0, o, mask_out ? &m : nullptr);
        *out = dup_string(o);
        if (mask_out) *mask_out = dup_string(m);
        return (int64_t)L;
    });
}

int kco_global(const char **simplitigs, uint64_t n, int k, int complements, char **out, char **mask_out) {
    if (n == 0) return -2;
    return dispatch(k, [&](auto kt) -> int {
        using K = decltype(kt);
        auto v = from_strings(simplitigs, n);
        auto hp = overlap_path<K>(v, k, complements != 0, false);
        std::string o, m;
        superstring_from_path<K>(hp, v, k, complements != 0, o, mask_out ? &m : nullptr);
        *out = dup_string(o);
        if (mask_out) *mask_out = dup_string(m);
        return 0;
    });
}

int kco_global_sparse(const uint64_t *words, uint64_t n, int k, int complements, int presort, char **out, char **mask_out) {
    if (n == 0) return -2;
    return dispatch(k, [&](auto kt) -> int {
        using K = decltype(kt);
        std::vector<K> v(n);
        for (u64 i = 0; i < n; ++i) v[i] = KT<K>::from_words(words + 4 * i);
        if (presort) partial_presort(v, k);
        std::vector<Simplitig> s(n, Simplitig(k));
        for (u64 i = 0; i < n; ++i) for (int j = 0; j < k; ++j) s[i][j] = (uint8_t)at_index(v[i], k, j);
        auto hp = overlap_path<K>(s, k, complements != 0, false);
        std::string o, m;
        superstring_from_path<K>(hp, s, k, complements != 0, o, mask_out ? &m : nullptr);
        *out = dup_string(o);
        if (mask_out) *mask_out = dup_string(m);
        return 0;
    });
}

// LowerBoundLength (lower_bound.h:7-21)
int64_t kco_lower_bound(const char **simplitigs, uint64_t n, int k, int complements) {
    return dispatch(k, [&](auto kt) -> int64_t {
        using K = decltype(kt);
        auto v = from_strings(simplitigs, n);
        auto hp = overlap_path<K>(v, k, complements != 0, true);
        u64 res = 0;
        for (auto &s : v) res += s.size() * 2 / (2 - (complements != 0));  // simplitig.size() is in bits there
        for (auto o : hp.overlaps) res -= o;
        return (int64_t)(res / (1 + (complements != 0)));
    });
}

// maskopt -t max-one / min-one on one masked superstring (masks.h:40-78, 239-261). Returns 0, -3 invalid chars.
int kco_maskopt(const char *ms, int k, int complements, int minimize, char **out) {
    std::string o;
    int rc = dispatch(k, [&](auto kt) -> int {
        using K = decltype(kt);
        return maskopt_ones<K>(std::string(ms), k, complements != 0, minimize != 0, o);
    });
    *out = dup_string(o);
    return rc;
}

void kco_reverse_complement(const uint64_t *in, int k, uint64_t *out) {
    dispatch(k, [&](auto kt) -> int { using K = decltype(kt); KT<K>::to_words(rev_comp(KT<K>::from_words(in), k), out); return 0; });
}
void kco_bit_prefix(const uint64_t *in, int k, int d, uint64_t *out) {
    dispatch(k, [&](auto kt) -> int { using K = decltype(kt); KT<K>::to_words(bit_prefix(KT<K>::from_words(in), k, d), out); return 0; });
}
void kco_bit_suffix(const uint64_t *in, int k, int d, uint64_t *out) {
    dispatch(k, [&](auto kt) -> int { using K = decltype(kt); KT<K>::to_words(bit_suffix(KT<K>::from_words(in), d), out); return 0; });
}
char kco_masked(char c, int m) { return masked(c, m != 0); }
int kco_nucleotide_to_int(int c) { return nuc((unsigned char)c); }
uint64_t kco_hash(const uint64_t *in, int k) {
    return dispatch(k, [&](auto kt) -> uint64_t { using K = decltype(kt); return ref_hash(KT<K>::from_words(in)); });
}
// key hash / edge priority of the canonical-order rule (shared spec with the HIP kernels; tests compare them).
uint64_t kco_key_hash(const uint64_t *in, int k) {
    return dispatch(k, [&](auto kt) -> uint64_t { using K = decltype(kt); return key_hash(KT<K>::from_words(in)); });
}

}  // extern "C"
