"""ctypes access to the CPU oracle (oracle/libkcoracle.so) and, where it was built, to the reference's own headers
compiled in place (oracle/_ref/libkcref.so).  TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only -- never by the product package."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libkcoracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libkcref.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "kcref")

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_vpp = ctypes.POINTER(ctypes.c_void_p)


def build_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "kc_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def words_for_k(k):
    return 1 if k < 32 else (2 if k < 64 else 4)


def _to_words(v):
    return [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]


def _from_words(buf, i):
    return buf[4 * i] | (buf[4 * i + 1] << 64) | (buf[4 * i + 2] << 128) | (buf[4 * i + 3] << 192)


class _Lib:
    """Common wrapper: the oracle (prefix kco_) and the reference shim (prefix kcref_) export the same entry points."""

    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.p = prefix
        self.free = getattr(self.lib, prefix + "free")
        self.free.argtypes = [ctypes.c_void_p]

    def fn(self, name):
        return getattr(self.lib, self.p + name)

    def _take(self, p):
        s = ctypes.string_at(p).decode()
        self.free(p)
        return s

    def kmer_set(self, seq, k, complements=True, case_sensitive=False):
        if isinstance(seq, str):
            seq = seq.encode()
        out = _u64p()
        f = self.fn("kmer_set")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_u64p)]
        n = f(seq, len(seq), k, int(complements), int(case_sensitive), ctypes.byref(out))
        res = [_from_words(out, i) for i in range(n)]
        self.free(ctypes.cast(out, ctypes.c_void_p))
        return res

    def read_kmers_count(self, path, k, complements=True, case_sensitive=False):
        f = self.fn("read_kmers_count")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        return f(path.encode(), k, int(complements), int(case_sensitive))

    def _strs(self, strs):
        return (ctypes.c_char_p * max(1, len(strs)))(*[s.encode() for s in strs])

    def overlap_path(self, strs, k, complements=True, lower_bound=False):
        n = len(strs)
        V = n * (2 if complements else 1)
        ef = (ctypes.c_uint64 * max(1, V))()
        ov = (ctypes.c_uint8 * max(1, V))()
        f = self.fn("overlap_path")
        f.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _u64p, _u8p]
        f(self._strs(strs), n, k, int(complements), int(lower_bound), ef, ov)
        return list(ef)[:V], list(ov)[:V]

    def superstring_from_path(self, strs, edge_from, overlaps, k, complements=True, want_mask=False):
        n = len(strs)
        V = len(edge_from)
        ef = (ctypes.c_uint64 * max(1, V))(*edge_from)
        ov = (ctypes.c_uint8 * max(1, V))(*overlaps)
        out, mask = ctypes.c_void_p(), ctypes.c_void_p()
        f = self.fn("superstring_from_path")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint64, _u64p, _u8p, ctypes.c_int, ctypes.c_int, _vpp, _vpp]
        L = f(self._strs(strs), n, ef, ov, k, int(complements), ctypes.byref(out), ctypes.byref(mask) if want_mask else None)
        o = self._take(out)
        m = self._take(mask).rstrip("\n") if want_mask else None
        return o, m, L

    def global_(self, strs, k, complements=True, want_mask=False):
        out, mask = ctypes.c_void_p(), ctypes.c_void_p()
        f = self.fn("global")
        f.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _vpp, _vpp]
        rc = f(self._strs(strs), len(strs), k, int(complements), ctypes.byref(out), ctypes.byref(mask) if want_mask else None)
        if rc:
            return rc, None, None
        return 0, self._take(out), (self._take(mask).rstrip("\n") if want_mask else None)

    def global_sparse(self, kmers, k, complements=True, presort=True, want_mask=False):
        n = len(kmers)
        buf = (ctypes.c_uint64 * max(1, 4 * n))()
        for i, v in enumerate(kmers):
            buf[4 * i:4 * i + 4] = _to_words(v)
        out, mask = ctypes.c_void_p(), ctypes.c_void_p()
        f = self.fn("global_sparse")
        f.argtypes = [_u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp, _vpp]
        rc = f(buf, n, k, int(complements), int(presort), ctypes.byref(out), ctypes.byref(mask) if want_mask else None)
        if rc:
            return rc, None, None
        return 0, self._take(out), (self._take(mask).rstrip("\n") if want_mask else None)

    def lower_bound(self, strs, k, complements=True):
        f = self.fn("lower_bound")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        return f(self._strs(strs), len(strs), k, int(complements))

    def reverse_complement(self, x, k):
        return self._unary("reverse_complement", x, k)

    def bit_prefix(self, x, k, d):
        return self._unary("bit_prefix", x, k, d)

    def bit_suffix(self, x, k, d):
        return self._unary("bit_suffix", x, k, d)

    def _unary(self, name, x, k, d=None):
        a = (ctypes.c_uint64 * 4)(*_to_words(x))
        o = (ctypes.c_uint64 * 4)()
        f = self.fn(name)
        if d is None:
            f.argtypes = [_u64p, ctypes.c_int, _u64p]
            f(a, k, o)
        else:
            f.argtypes = [_u64p, ctypes.c_int, ctypes.c_int, _u64p]
            f(a, k, d, o)
        return _from_words(o, 0)

    def masked(self, c, m):
        f = self.fn("masked")
        f.restype = ctypes.c_char
        f.argtypes = [ctypes.c_char, ctypes.c_int]
        return f(c.encode(), int(m)).decode()

    def nucleotide_to_int(self, c):
        return self.fn("nucleotide_to_int")(ord(c))

    def hash(self, x, k):
        a = (ctypes.c_uint64 * 4)(*_to_words(x))
        f = self.fn("hash")
        f.restype = ctypes.c_uint64
        f.argtypes = [_u64p, ctypes.c_int]
        return f(a, k)


class Oracle(_Lib):
    def __init__(self):
        build_oracle()
        super().__init__(ORACLE_SO, "kco_")

    def simplitigs_of_sequence(self, seq, k, complements=True, mode=1):
        if isinstance(seq, str):
            seq = seq.encode()
        f = self.fn("simplitigs_of_sequence")
        f.restype = ctypes.c_void_p
        f.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        return [s for s in self._take(f(seq, len(seq), k, int(complements), mode)).split("\n") if s]

    def simplitigs_of_kmers(self, kmers, k, complements=True, mode=1):
        n = len(kmers)
        buf = (ctypes.c_uint64 * max(1, 4 * n))()
        for i, v in enumerate(kmers):
            buf[4 * i:4 * i + 4] = _to_words(v)
        f = self.fn("simplitigs_of_kmers")
        f.restype = ctypes.c_void_p
        f.argtypes = [_u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        return [s for s in self._take(f(buf, n, k, int(complements), mode)).split("\n") if s]

    def compute_file(self, path, k, complements=True, assume_simplitigs=False, mode=1, want_mask=True):
        """-> (rc, out, mask, stats[8]) ; mode 0 = reference order, 1 = canonical order"""
        out, mask = ctypes.c_void_p(), ctypes.c_void_p()
        st = (ctypes.c_uint64 * 8)()
        f = self.fn("compute_file")
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp, _vpp, _u64p]
        rc = f(path.encode(), k, int(complements), int(assume_simplitigs), mode, ctypes.byref(out), ctypes.byref(mask) if want_mask else None, st)
        if rc:
            return rc, None, None, None
        return 0, self._take(out), (self._take(mask) if want_mask else None), list(st)

    def compute_file_z(self, path, k, complements=True, mode=1, min_frequency=2, want_mask=True):
        """compute -z: -> (rc, out, mask, stats[8])"""
        out, mask = ctypes.c_void_p(), ctypes.c_void_p()
        st = (ctypes.c_uint64 * 8)()
        f = self.fn("compute_file_z")
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp, _vpp, _u64p]
        rc = f(path.encode(), k, int(complements), mode, min_frequency, ctypes.byref(out), ctypes.byref(mask) if want_mask else None, st)
        if rc:
            return rc, None, None, None
        return 0, self._take(out), (self._take(mask) if want_mask else None), list(st)

    def file_kmer_set_z(self, path, k, complements=True, min_frequency=2):
        out = _u64p()
        f = self.fn("file_kmer_set_z")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_u64p)]
        n = f(path.encode(), k, int(complements), min_frequency, ctypes.byref(out))
        if n < 0:
            return None
        res = [_from_words(out, i) for i in range(n)]
        self.free(ctypes.cast(out, ctypes.c_void_p))
        return res

    def file_kmer_set(self, path, k, complements=True, case_sensitive=False):
        out = _u64p()
        f = self.fn("file_kmer_set")
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_u64p)]
        n = f(path.encode(), k, int(complements), int(case_sensitive), ctypes.byref(out))
        if n < 0:
            return None
        res = [_from_words(out, i) for i in range(n)]
        self.free(ctypes.cast(out, ctypes.c_void_p))
        return res

    def parse_records(self, path):
        f = self.fn("parse_records")
        f.restype = ctypes.c_void_p
        f.argtypes = [ctypes.c_char_p]
        p = f(path.encode())
        if not p:
            return None
        return [tuple(line.split("\t")) for line in self._take(p).split("\n") if line]

    def maskopt(self, ms, k, complements=True, minimize=False):
        out = ctypes.c_void_p()
        f = self.fn("maskopt")
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]
        rc = f(ms.encode(), k, int(complements), int(minimize), ctypes.byref(out))
        return rc, self._take(out)

    def key_hash(self, x, k):
        a = (ctypes.c_uint64 * 4)(*_to_words(x))
        f = self.fn("key_hash")
        f.restype = ctypes.c_uint64
        f.argtypes = [_u64p, ctypes.c_int]
        return f(a, k)


class Reference(_Lib):
    """The reference's own headers compiled in place (only where oracle/_ref was built)."""

    def __init__(self):
        super().__init__(REF_SO, "kcref_")

    def simplitigs_of_sequence(self, seq, k, complements=True):
        if isinstance(seq, str):
            seq = seq.encode()
        f = self.fn("simplitigs_of_sequence")
        f.restype = ctypes.c_void_p
        f.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        return [s for s in self._take(f(seq, len(seq), k, int(complements))).split("\n") if s]

    def simplitigs_of_kmers(self, kmers, k, complements=True):
        n = len(kmers)
        buf = (ctypes.c_uint64 * max(1, 4 * n))()
        for i, v in enumerate(kmers):
            buf[4 * i:4 * i + 4] = _to_words(v)
        f = self.fn("simplitigs_of_kmers")
        f.restype = ctypes.c_void_p
        f.argtypes = [_u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
        return [s for s in self._take(f(buf, n, k, int(complements))).split("\n") if s]

    def compute_file(self, path, k, complements=True, assume_simplitigs=False, want_mask=True, tmpdir="/tmp"):
        st = (ctypes.c_uint64 * 8)()
        o = os.path.join(tmpdir, "kcref_out_%d.txt" % os.getpid())
        m = os.path.join(tmpdir, "kcref_mask_%d.txt" % os.getpid())
        f = self.fn("compute_file")
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, _u64p]
        rc = f(path.encode(), k, int(complements), int(assume_simplitigs), o.encode(), m.encode() if want_mask else b"", st)
        if rc:
            return rc, None, None, None
        out = open(o).read().rstrip("\n")
        mask = open(m).read().rstrip("\n") if want_mask else None
        for p in (o, m):
            if os.path.exists(p):
                os.remove(p)
        return 0, out, mask, list(st)


def _ref_z(self, path, k, complements, min_frequency, want_mask=True, tmpdir="/tmp"):
    st = (ctypes.c_uint64 * 8)()
    o = os.path.join(tmpdir, "kcref_zout_%d.txt" % os.getpid())
    m = os.path.join(tmpdir, "kcref_zmask_%d.txt" % os.getpid())
    f = self.fn("compute_file_z")
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, _u64p]
    rc = f(path.encode(), k, int(complements), min_frequency, o.encode(), m.encode() if want_mask else b"", st)
    if rc:
        return rc, None, None, None
    out = open(o).read().rstrip("\n")
    mask = open(m).read().rstrip("\n") if want_mask else None
    for pth in (o, m):
        if os.path.exists(pth):
            os.remove(pth)
    return 0, out, mask, list(st)


def _ref_zcount(self, path, k, complements, min_frequency):
    f = self.fn("read_kmers_filtered_count")
    f.restype = ctypes.c_int64
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return f(path.encode(), k, int(complements), min_frequency)


def _ref_split_ms(self, path):
    f = self.fn("split_ms")
    f.argtypes = [ctypes.c_char_p, _vpp, _vpp]
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    if f(path.encode(), ctypes.byref(a), ctypes.byref(b)):
        return None
    return self._take(a), self._take(b)


def _ref_join_ms(self, superstring_text, mask_text):
    f = self.fn("join_ms")
    f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, _vpp]
    a = ctypes.c_void_p()
    f(superstring_text.encode(), mask_text.encode(), ctypes.byref(a))
    return self._take(a)


def _ref_convert(name):
    def call(self, path, k):
        f = self.fn(name)
        f.argtypes = [ctypes.c_char_p, ctypes.c_int, _vpp]
        a = ctypes.c_void_p()
        if f(path.encode(), k, ctypes.byref(a)):
            return None
        return self._take(a)
    return call


Reference.split_ms = _ref_split_ms
Reference.join_ms = _ref_join_ms
Reference.ms_to_spss = _ref_convert("ms_to_spss")
Reference.spss_to_ms = _ref_convert("spss_to_ms")
Reference.compute_file_z = _ref_z
Reference.read_kmers_filtered_count = _ref_zcount


def have_reference():
    return os.path.exists(REF_SO)


# ---- pure-python verifier (the jellyfish-free restatement of verify.py:8-61) ----
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def kmers_of_masked_superstring(ms, k, complements=True):
    """distinct (canonical) k-mers starting at upper-case positions; also returns the number of ones."""
    up = ms.upper()
    out = set()
    ones = 0
    for i in range(len(ms) - k + 1):
        if ms[i] <= "Z":
            ones += 1
            x = up[i:i + k]
            if complements:
                r = "".join(_COMP[c] for c in reversed(x))
                if r < x:
                    x = r
            out.add(x)
    return out, ones


def encode_kmer(s):
    v = 0
    for c in s:
        v = (v << 2) | "ACGT".index(c)
    return v
