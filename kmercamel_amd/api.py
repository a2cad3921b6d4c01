"""ctypes binding of libkmercamel_amd.so (include/kmercamel_amd.h) plus the host-side mirror of the reference's
operator interface for the BIGREEDY path.

Nothing here computes on the CPU: every operator goes through the C ABI into the HIP kernels and raises
`KcError` if the library or a GPU is missing.  Function names follow the reference
(`/root/reference/src/{parser,simplitigs,global}.h`): ReadKMers, get_simplitigs, simplitigs_from_fasta,
OverlapHamiltonianPath, SuperstringFromPath, Global, LowerBoundLength, Optimize (max-one/min-one).
"""
import ctypes
import gzip
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KMERCAMEL_AMD_LIB: the tools' A/B runs point this at another in-tree build, e.g. `make experiments` -> libkmercamel_amd_exp.so)
LIB_PATH = os.environ.get("KMERCAMEL_AMD_LIB") or os.path.join(_HERE, "libkmercamel_amd.so")

KC_OK = 0
KC_E_NO_DEVICE, KC_E_EMPTY, KC_E_INVALID_CHAR, KC_E_ARG, KC_E_CAPACITY, KC_E_IO, KC_E_NO_KMERS, KC_E_TRANSPORT, KC_E_UNSUPPORTED_SHARDED = -1, -2, -3, -4, -5, -6, -7, -8, -9
FLAG_COMPLEMENTS, FLAG_SIMPLITIGS_INPUT, FLAG_WANT_MASK, FLAG_INPUT_ON_DEVICE = 1, 2, 4, 8

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_vp = ctypes.c_void_p
_vpp = ctypes.POINTER(ctypes.c_void_p)

# every symbol include/kmercamel_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kc_ctx_create": (ctypes.c_int, [ctypes.c_int, _vpp]),
    "kc_ctx_destroy": (None, [_vp]),
    "kc_last_error": (ctypes.c_char_p, [_vp]),
    "kc_ctx_stream": (_vp, [_vp]),
    "kc_ctx_synchronize": (ctypes.c_int, [_vp]),
    "kc_ctx_stats": (ctypes.c_int, [_vp, _u64p]),
    "kc_set_create": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, _vpp]),
    "kc_set_destroy": (None, [_vp]),
    "kc_set_add_sequence": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int]),
    "kc_set_add_sequence_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_int]),
    "kc_set_enable_counting": (ctypes.c_int, [_vp]),
    "kc_set_filter_min_frequency": (ctypes.c_int, [_vp, ctypes.c_int]),
    "kc_set_size": (ctypes.c_int, [_vp, _u64p]),
    "kc_set_export": (ctypes.c_int, [_vp, _u64p]),
    "kc_set_contains": (ctypes.c_int, [_vp, _u64p, ctypes.c_uint64, _u8p]),
    "kc_set_erase": (ctypes.c_int, [_vp, _u64p, ctypes.c_uint64, _u64p]),
    "kc_simplitigs_build": (ctypes.c_int, [_vp, _vpp]),
    "kc_simplitigs_from_strings": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_char_p, _u64p, ctypes.c_uint64, _vpp]),
    "kc_simplitigs_destroy": (None, [_vp]),
    "kc_simplitigs_count": (ctypes.c_int, [_vp, _u64p, _u64p]),
    "kc_simplitigs_export": (ctypes.c_int, [_vp, ctypes.c_char_p, _u64p]),
    "kc_simplitigs_to_sparse": (ctypes.c_int, [_vp, _vpp]),
    "kc_overlap_path": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _u64p, _u8p]),
    "kc_superstring_from_path": (ctypes.c_int, [_vp, _u64p, _u8p, ctypes.c_int, _vp, _vpp, _vpp, _u64p]),
    "kc_global": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _vpp, _vpp, _u64p]),
    "kc_lower_bound": (ctypes.c_int, [_vp, ctypes.c_int, _u64p]),
    "kc_free": (None, [_vp]),
    "kc_compute": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, _u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, _vpp]),
    "kc_result_stats": (ctypes.c_int, [_vp, _u64p]),
    "kc_result_stats_ext": (ctypes.c_int, [_vp, _u64p, ctypes.c_int]),
    "kc_result_device_ptrs": (ctypes.c_int, [_vp, _vpp, _vpp]),
    "kc_result_copy": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_char_p]),
    "kc_result_destroy": (None, [_vp]),
    "kc_comm_create_self": (ctypes.c_int, [_vp, _vpp]),
    "kc_comm_create_callbacks": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vpp]),
    "kc_rccl_unique_id": (ctypes.c_int, [_vp]),
    "kc_comm_create_rccl": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vpp]),
    "kc_comm_destroy": (None, [_vp]),
    "kc_comm_stats": (ctypes.c_int, [_vp, _u64p]),
    "kc_thread_group_create": (ctypes.c_int, [ctypes.c_int, _vpp]),
    "kc_thread_group_destroy": (None, [_vp]),
    "kc_comm_create_threads": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vpp]),
    "kc_device_count": (ctypes.c_int, []),
    "kc_shard_geometry": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _u64p]),
    "kc_shard_owner_of": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, _u64p, ctypes.c_uint64, _u8p]),
    "kc_compute_sharded": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, _vpp]),
    "kc_compute_sharded_simplitigs": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint64, _u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32, _vpp]),
    "kc_result_slice": (ctypes.c_int, [_vp, _u64p, _u64p]),
    "kc_read_fasta_device": (ctypes.c_int, [_vp, ctypes.c_char_p, _vpp, _u64p, ctypes.POINTER(ctypes.c_int)]),
    "kc_read_fasta_device_range": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _vpp, _u64p, ctypes.POINTER(ctypes.c_int)]),
    "kc_device_free": (None, [_vp, _vp]),
    "kc_result_write_fd": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "kc_result_write_mem": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "kc_maskopt_ones": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    "kc_maskopt_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vpp]),
    "kc_version": (ctypes.c_char_p, []),
}

STAT_NAMES = ["n_kmers", "n_simplitigs", "length", "sparse_branch", "n_vertices", "matching_rounds", "us_insert_phase",
              "us_simplitig_phase", "us_overlap_phase", "us_emit_phase", "us_probe_kernel", "us_insert_kernel", "table_capacity",
              "overlap_repairs", "us_total", "reserved"]


class KcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kmercamel_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the C-ABI library; fails loudly if it was not built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KcError(KC_E_NO_DEVICE, "libkmercamel_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI symbol missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _host_bytes(seq):
    """(void pointer, length, keep-alive object) of a str / bytes / bytearray / contiguous numpy uint8 array, without copying arrays"""
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        b = bytes(seq)
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), len(b), b
    import numpy as np
    a = np.ascontiguousarray(seq, dtype=np.uint8)
    return ctypes.c_void_p(a.ctypes.data), int(a.size), a


def words_for_k(k):
    return 1 if k < 32 else (2 if k < 64 else 4)


class Context:
    """One per GPU: private HIP stream + device memory pool (kc_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        h = ctypes.c_void_p()
        rc = self.lib.kc_ctx_create(device, ctypes.byref(h))
        if rc != KC_OK:
            raise KcError(rc, "kc_ctx_create failed (no HIP device?)")
        self.h = h
        self._children = []

    def check(self, rc):
        if rc != KC_OK:
            raise KcError(rc, (self.lib.kc_last_error(self.h) or b"").decode())

    def close(self):
        """Destroys the context. Objects created from it (sets, simplitigs, results) must be closed first; any that are
        still alive are invalidated here so that their finalisers do not touch freed device state."""
        if self.h:
            for ref in list(self._children):
                obj = ref()
                if obj is not None:
                    obj.close()
            self._children = []
            self.lib.kc_ctx_destroy(self.h)
            self.h = None

    def _adopt(self, obj):
        import weakref
        self._children.append(weakref.ref(obj))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerSet:
    """kh_S64/S128/S256 + kmer_dictNN_t (khash_utils.h:28-95) on the device."""

    def __init__(self, ctx, k, complements=True, capacity_hint=0):
        self.ctx, self.k, self.complements = ctx, k, bool(complements)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.kc_set_create(ctx.h, k, int(self.complements), capacity_hint, ctypes.byref(h)))
        self.h = h
        ctx._adopt(self)

    def AddKMers(self, seq, case_sensitive=False):  # parser.h:22-49
        ptr, n, _keep = _host_bytes(seq)
        self.ctx.check(self.ctx.lib.kc_set_add_sequence(self.h, ctypes.cast(ptr, ctypes.c_char_p), n, int(case_sensitive)))

    def add_device(self, ptr, length, case_sensitive=False):
        self.ctx.check(self.ctx.lib.kc_set_add_sequence_device(self.h, ctypes.c_void_p(ptr), length, int(case_sensitive)))

    def size(self):  # kh_size
        n = ctypes.c_uint64()
        self.ctx.check(self.ctx.lib.kc_set_size(self.h, ctypes.byref(n)))
        return n.value

    def keys(self):
        """kMersToVec (khash_utils.h:131-141): python ints, unsorted."""
        n, W = self.size(), words_for_k(self.k)
        buf = (ctypes.c_uint64 * max(1, n * W))()
        self.ctx.check(self.ctx.lib.kc_set_export(self.h, buf))
        out = []
        for i in range(n):
            v = 0
            for j in range(W):
                v |= buf[i * W + j] << (64 * j)
            out.append(v)
        return out

    def contains(self, keys):
        W = words_for_k(self.k)
        n = len(keys)
        buf = (ctypes.c_uint64 * max(1, n * W))()
        for i, v in enumerate(keys):
            for j in range(W):
                buf[i * W + j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        out = (ctypes.c_uint8 * max(1, n))()
        self.ctx.check(self.ctx.lib.kc_set_contains(self.h, buf, n, out))
        return [bool(out[i]) for i in range(n)]

    def close(self):
        if self.h:
            self.ctx.lib.kc_set_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Simplitigs:
    """std::vector<simplitig_t> (simplitigs.h:11) as a packed device-resident collection."""

    def __init__(self, ctx, h, k):
        self.ctx, self.h, self.k = ctx, h, k
        ctx._adopt(self)

    def count(self):
        n, t = ctypes.c_uint64(), ctypes.c_uint64()
        self.ctx.check(self.ctx.lib.kc_simplitigs_count(self.h, ctypes.byref(n), ctypes.byref(t)))
        return n.value, t.value

    def strings(self):
        n, t = self.count()
        buf = ctypes.create_string_buffer(max(1, t))
        off = (ctypes.c_uint64 * (n + 1))()
        self.ctx.check(self.ctx.lib.kc_simplitigs_export(self.h, buf, off))
        raw = buf.raw
        return [raw[off[i]:off[i + 1]].decode() for i in range(n)]

    def to_sparse(self):
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.kc_simplitigs_to_sparse(self.h, ctypes.byref(h)))
        return Simplitigs(self.ctx, h, self.k)

    def close(self):
        if self.h:
            self.ctx.lib.kc_simplitigs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _take_string(lib, p):
    if not p:
        return None
    s = ctypes.string_at(p).decode()
    lib.kc_free(p)
    return s


# ---------------------------------------------------------------------------------------------------------
# host-side record reader with kseq semantics (kseq.h:183-224; gz transparently like parser.h:88-101)
def read_records(path):
    """[(name, comment, sequence)] of a FASTA/FASTQ(.gz) file, multi-line records joined."""
    if path == "-":
        import sys
        data = sys.stdin.buffer.read()
    else:
        with open(path, "rb") as f:
            data = f.read()
    if data[:2] == b"\x1f\x8b":
        data = gzip.decompress(data)
    out = []
    n, p, last_char = len(data), 0, 0

    def take_line(dst):
        nonlocal p
        e = data.find(b"\n", p)
        if e < 0:
            e = n
        dst += data[p:e]
        if len(dst) > 1 and dst[-1:] == b"\r":
            del dst[-1]
        p = e + 1 if e < n else n

    while True:
        if last_char == 0:
            while p < n and data[p] not in (0x3E, 0x40):
                p += 1
            if p >= n:
                break
            last_char = data[p]
            p += 1
        if p >= n:
            break
        q = p
        while q < n and data[q] not in b" \t\n\r\x0b\x0c":
            q += 1
        name = data[p:q]
        comment = bytearray()
        if q < n:
            delim = data[q]
            p = q + 1
            if delim != 0x0A:
                take_line(comment)
        else:
            p = n
        seq = bytearray()
        c = -1
        while p < n:
            c = data[p]
            p += 1
            if c in (0x3E, 0x2B, 0x40):
                break
            if c == 0x0A:
                c = -1
                continue
            seq.append(c)
            take_line(seq)
            c = -1
        if c in (0x3E, 0x40):
            last_char = c
        if c != 0x2B:
            out.append((name.decode(), bytes(comment).decode(), bytes(seq)))
            if c == -1:
                break
            continue
        e = data.find(b"\n", p)
        if e < 0:
            break
        p = e + 1
        qual = bytearray()
        while p < n:
            take_line(qual)
            if not len(qual) < len(seq):
                break
        last_char = 0
        if len(qual) != len(seq):
            break
        out.append((name.decode(), bytes(comment).decode(), bytes(seq)))
    return out


def ReadKMers(ctx, path, k, complements=True, case_sensitive=False):
    """parser.h:106-118: k-mer set of every record of a FASTA file."""
    recs = read_records(path)
    joined = b"\n".join(r[2] for r in recs)
    s = KmerSet(ctx, k, complements, capacity_hint=len(joined))
    if joined:
        s.AddKMers(joined, case_sensitive)
    return s


def ReadKMersFiltered(ctx, path, k, complements=True, min_frequency=1):
    """parser.h:122-141: k-mers seen at least min_frequency times (-z)."""
    recs = read_records(path)
    joined = b"\n".join(r[2] for r in recs)
    s = KmerSet(ctx, k, complements, capacity_hint=len(joined))
    ctx.check(ctx.lib.kc_set_enable_counting(s.h))
    if joined:
        s.AddKMers(joined)
    ctx.check(ctx.lib.kc_set_filter_min_frequency(s.h, min_frequency))
    return s


def get_simplitigs(kmer_set):
    """simplitigs.h:191-205 (canonical order; the set is not consumed)."""
    ctx = kmer_set.ctx
    h = ctypes.c_void_p()
    ctx.check(ctx.lib.kc_simplitigs_build(kmer_set.h, ctypes.byref(h)))
    return Simplitigs(ctx, h, kmer_set.k)


def simplitigs_from_strings(ctx, strings, k):
    """simplitig_from_string (simplitigs.h:78-87) for each string, order kept."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
    off = (ctypes.c_uint64 * (len(bs) + 1))()
    o = 0
    for i, b in enumerate(bs):
        off[i] = o
        o += len(b)
    off[len(bs)] = o
    h = ctypes.c_void_p()
    ctx.check(ctx.lib.kc_simplitigs_from_strings(ctx.h, k, b"".join(bs), off, len(bs), ctypes.byref(h)))
    return Simplitigs(ctx, h, k)


def simplitigs_from_fasta(ctx, path, k):
    """simplitigs.h:89-103"""
    return simplitigs_from_strings(ctx, [r[2] for r in read_records(path)], k)


def OverlapHamiltonianPath(simplitigs, complements=True, lower_bound=False):
    """global.h:43-133 -> (edgeFrom list with None for -1, overlaps list with None for 0xFF)."""
    ctx = simplitigs.ctx
    n, _ = simplitigs.count()
    V = n * (2 if complements else 1)
    ef = (ctypes.c_uint64 * max(1, V))()
    ov = (ctypes.c_uint8 * max(1, V))()
    ctx.check(ctx.lib.kc_overlap_path(simplitigs.h, int(bool(complements)), int(bool(lower_bound)), ef, ov))
    return list(ef)[:V], list(ov)[:V]


def SuperstringFromPath(simplitigs, edge_from, overlaps, complements=True, kmer_set=None, want_mask=False):
    """global.h:149-210 -> (superstring, max-one mask or None, length)."""
    ctx = simplitigs.ctx
    V = len(edge_from)
    ef = (ctypes.c_uint64 * max(1, V))(*edge_from)
    ov = (ctypes.c_uint8 * max(1, V))(*overlaps)
    out, mask, L = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
    ctx.check(ctx.lib.kc_superstring_from_path(simplitigs.h, ef, ov, int(bool(complements)), kmer_set.h if kmer_set else None,
                                               ctypes.byref(out), ctypes.byref(mask) if want_mask else None, ctypes.byref(L)))
    return _take_string(ctx.lib, out), _take_string(ctx.lib, mask) if want_mask else None, L.value


def Global(simplitigs, complements=True, want_mask=False, kmer_set=None):
    """global.h:217-226 -> (superstring, mask or None, length); raises KcError(KC_E_EMPTY) on empty input."""
    ctx = simplitigs.ctx
    out, mask, L = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
    ctx.check(ctx.lib.kc_global(simplitigs.h, int(bool(complements)), kmer_set.h if kmer_set else None, int(bool(want_mask)),
                                ctypes.byref(out), ctypes.byref(mask) if want_mask else None, ctypes.byref(L)))
    return _take_string(ctx.lib, out), _take_string(ctx.lib, mask) if want_mask else None, L.value


def LowerBoundLength(simplitigs, complements=True):
    """lower_bound.h:7-21"""
    ctx = simplitigs.ctx
    L = ctypes.c_uint64()
    ctx.check(ctx.lib.kc_lower_bound(simplitigs.h, int(bool(complements)), ctypes.byref(L)))
    return L.value


def Optimize(ctx, masked_superstring, k, complements=True, algorithm="max-one"):
    """masks.h:239-261 for max-one / min-one on the sequence line of a masked superstring."""
    if algorithm not in ("max-one", "min-one"):
        raise KcError(KC_E_ARG, "Algorithm '%s' not recognized." % algorithm)
    ms = masked_superstring.encode() if isinstance(masked_superstring, str) else bytes(masked_superstring)
    out = ctypes.create_string_buffer(len(ms) + 1)
    ctx.check(ctx.lib.kc_maskopt_ones(ctx.h, ms, len(ms), k, int(bool(complements)), int(algorithm == "min-one"), out))
    return out.raw[:len(ms)].decode()


class ComputeResult:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._adopt(self)
        st = (ctypes.c_uint64 * 19)()
        ctx.check(ctx.lib.kc_result_stats_ext(h, st, 19))
        self.stats = dict(zip(STAT_NAMES + ["cycles_cut", "n_kmers_rank", "queries_sent"], list(st)))
        self.length = self.stats["length"]

    def device_ptrs(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.ctx.check(self.ctx.lib.kc_result_device_ptrs(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def strings(self, want_mask=False):
        """(-o line, -M line) held by this result: the whole superstring, or -- hash-sharded runs -- this rank's slice of it"""
        n = self.slice()[1]
        out = ctypes.create_string_buffer(n + 1)
        mask = ctypes.create_string_buffer(n + 1) if want_mask else None
        self.ctx.check(self.ctx.lib.kc_result_copy(self.h, out, mask))
        return out.raw[:n].decode(), (mask.raw[:n].decode() if want_mask else None)

    def slice(self):
        """(offset, length) of the part of the superstring this result holds (hash-sharded runs: this rank's slice)"""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self.ctx.check(self.ctx.lib.kc_result_slice(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def array(self):
        """the masked superstring (or this rank's slice of it) as a numpy uint8 array -- no Python string in between"""
        import numpy as np
        n = self.slice()[1]
        out = np.empty(n + 1, dtype=np.uint8)
        self.ctx.check(self.ctx.lib.kc_result_copy(self.h, out.ctypes.data_as(ctypes.c_char_p), None))
        return out[:n]

    def close(self):
        if self.h:
            self.ctx.lib.kc_result_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def compute(ctx, seq, k, complements=True, want_mask=False, simplitig_offsets=None, device_ptr=None, length=None, min_frequency=1):
    """Fused device-resident greedy branch of kmercamel<>() (main.cpp:122-212).

    seq: bytes of the record sequences joined by '\\n' (or, with device_ptr/length, already on the GPU);
    simplitig_offsets: for -S, the n+1 record offsets into seq (records then are NOT separated)."""
    flags = (FLAG_COMPLEMENTS if complements else 0) | (FLAG_WANT_MASK if want_mask else 0) | ((min_frequency & 0xFF) << 8)
    off, nrec = None, 0
    if simplitig_offsets is not None:
        flags |= FLAG_SIMPLITIGS_INPUT
        nrec = len(simplitig_offsets) - 1
        off = (ctypes.c_uint64 * len(simplitig_offsets))(*simplitig_offsets)
    h = ctypes.c_void_p()
    if device_ptr is not None:
        flags |= FLAG_INPUT_ON_DEVICE
        ctx.check(ctx.lib.kc_compute(ctx.h, ctypes.c_void_p(device_ptr), length, off, nrec, k, flags, ctypes.byref(h)))
    else:
        ptr, n, _keep = _host_bytes(seq)
        ctx.check(ctx.lib.kc_compute(ctx.h, ptr, n, off, nrec, k, flags, ctypes.byref(h)))
    return ComputeResult(ctx, h)


# ---- hash-sharded multi-GPU path: one process per GPU -------------------------------------------------
_A2A_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _u64p, ctypes.c_void_p, _u64p)
_AG_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)


def host_transport(dist, group=None):
    """The two host-buffer collectives kc_comm_create_callbacks asks for, over a torch.distributed process group (gloo):
    (alltoallv(user, send, send_bytes, recv, recv_bytes), allgather(user, in, out, bytes_per_rank)), raw pointers in, 0 = ok.
    `send` holds the bytes for rank 0, 1, ... back to back; `recv` receives rank 0's, 1's, ... bytes back to back."""
    import torch
    world = dist.get_world_size(group)

    def view(ptr, nbytes):
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8)
        return torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

    def a2a(user, send, send_bytes, recv, recv_bytes):
        try:
            sb = [int(send_bytes[i]) for i in range(world)]
            rb = [int(recv_bytes[i]) for i in range(world)]
            dist.all_to_all_single(view(recv, sum(rb)), view(send, sum(sb)), output_split_sizes=rb, input_split_sizes=sb, group=group)
            return 0
        except Exception as e:  # noqa: BLE001 -- an exception must not unwind through the C caller
            print("kmercamel_amd: all-to-all callback failed: %r" % (e,), flush=True)
            return 1

    def ag(user, inp, out, nbytes):
        try:
            n = int(nbytes)
            dist.all_gather_into_tensor(view(out, n * world), view(inp, n), group=group)
            return 0
        except Exception as e:  # noqa: BLE001
            print("kmercamel_amd: all-gather callback failed: %r" % (e,), flush=True)
            return 1

    return a2a, ag


def shard_geometry(k, total_occurrences, world, rank):
    """kc_shard_geometry (host side, no GPU): dict of the table partition for `world` ranks"""
    lib = load()
    out = (ctypes.c_uint64 * 6)()
    rc = lib.kc_shard_geometry(k, total_occurrences, world, rank, out)
    if rc != KC_OK:
        raise KcError(rc, "kc_shard_geometry")
    return dict(zip(("nblocks", "bshift", "blocks_per_rank", "first_slot", "slots_per_rank", "minimizer_length"), [int(x) for x in out]))


def shard_owner_of(k, total_occurrences, world, keys):
    """kc_shard_owner_of (host side, no GPU): owner rank of canonical k-mers given as a numpy uint64 array of shape (n, W)"""
    import numpy as np
    lib = load()
    a = np.ascontiguousarray(keys, dtype=np.uint64)
    n = a.shape[0] if a.ndim == 2 else a.size // words_for_k(k)
    out = np.empty(max(1, n), dtype=np.uint8)
    rc = lib.kc_shard_owner_of(k, total_occurrences, world, a.ctypes.data_as(_u64p), n, out.ctypes.data_as(_u8p))
    if rc != KC_OK:
        raise KcError(rc, "kc_shard_owner_of")
    return out[:n]


class Comm:
    """Rank-to-rank transport of the hash-sharded path (kc_comm).

    Comm.self_(ctx)             one rank, no transport;
    Comm.rccl(ctx, dist)        RCCL over xGMI inside the library (grouped ncclSend/ncclRecv); torch.distributed is only used to
                                hand rank 0's unique id to the other ranks;
    Comm.torch_host(ctx, dist)  host-staged through a torch.distributed process group (gloo): tests, boxes without xGMI."""

    def __init__(self, ctx, h, rank, world, keep=None):
        self.ctx, self.h, self.rank, self.world, self._keep = ctx, h, rank, world, keep

    @classmethod
    def self_(cls, ctx):
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.kc_comm_create_self(ctx.h, ctypes.byref(h)))
        return cls(ctx, h, 0, 1)

    @classmethod
    def rccl(cls, ctx, dist):
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = (ctypes.c_uint8 * 128)()
        if rank == 0:
            ctx.check(ctx.lib.kc_rccl_unique_id(buf))
        box = [bytes(buf)]
        dist.broadcast_object_list(box, src=0)
        idb = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.kc_comm_create_rccl(ctx.h, rank, world, idb, ctypes.byref(h)))
        return cls(ctx, h, rank, world)

    @classmethod
    def torch_host(cls, ctx, dist, group=None):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        a2a, ag = host_transport(dist, group)
        fa, fg = _A2A_FN(a2a), _AG_FN(ag)
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.kc_comm_create_callbacks(ctx.h, rank, world, ctypes.cast(fa, ctypes.c_void_p), ctypes.cast(fg, ctypes.c_void_p), None,
                                                   ctypes.byref(h)))
        return cls(ctx, h, rank, world, keep=(fa, fg))

    @classmethod
    def threads(cls, ctx, group, rank, world):
        """rank `rank` of an in-process thread group (kc_thread_group_create(world)): the transport of `kmercamel compute --gpus N`"""
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.kc_comm_create_threads(ctx.h, group, rank, ctypes.byref(h)))
        return cls(ctx, h, rank, world)

    def stats(self):
        st = (ctypes.c_uint64 * 8)()
        self.ctx.check(self.ctx.lib.kc_comm_stats(self.h, st))
        return {"bytes_sent": st[0], "bytes_received": st[1], "calls": st[2], "us_in_collectives": st[3], "self_sendrecv_calls": st[4]}

    def close(self):
        if self.h:
            self.ctx.lib.kc_comm_destroy(self.h)
            self.h = None


def compute_sharded(ctx, comm, seq, k, complements=True, device_ptr=None, length=None, want_mask=False, min_frequency=1):
    """kc_compute_sharded: collective over comm's ranks; `seq` = THIS rank's records joined by '\\n' (may be empty).
    The ComputeResult holds this rank's slice of the one masked superstring (see .slice(), .array(), .strings(want_mask))."""
    flags = (FLAG_COMPLEMENTS if complements else 0) | (FLAG_WANT_MASK if want_mask else 0) | ((min_frequency & 0xff) << 8 if min_frequency > 1 else 0)
    h = ctypes.c_void_p()
    if device_ptr is not None:
        flags |= FLAG_INPUT_ON_DEVICE
        ctx.check(ctx.lib.kc_compute_sharded(ctx.h, comm.h, ctypes.c_void_p(device_ptr), length, k, flags, ctypes.byref(h)))
    else:
        ptr, n, _keep = _host_bytes(seq)
        ctx.check(ctx.lib.kc_compute_sharded(ctx.h, comm.h, ptr, n, k, flags, ctypes.byref(h)))
    return ComputeResult(ctx, h)


def compute_sharded_simplitigs(ctx, comm, records, k, complements=True, want_mask=False):
    """kc_compute_sharded_simplitigs (-S): `records` = THIS rank's simplitigs (list of bytes); collective over comm's ranks."""
    offs, o = [], 0
    for r in records:
        offs.append(o)
        o += len(r)
    offs.append(o)
    arr = (ctypes.c_uint64 * len(offs))(*offs)
    ptr, n, _keep = _host_bytes(b"".join(records))
    h = ctypes.c_void_p()
    ctx.check(ctx.lib.kc_compute_sharded_simplitigs(ctx.h, comm.h, ptr, n, arr, len(records), k, (FLAG_COMPLEMENTS if complements else 0) | (FLAG_WANT_MASK if want_mask else 0),
                                                    ctypes.byref(h)))
    return ComputeResult(ctx, h)


def compute_file(ctx, path, k, complements=True, want_mask=False, assume_simplitigs=False, min_frequency=1):
    """`kmercamel compute -k K [-u] [-S] [-M f] path` on the sequence lines (no FASTA header)."""
    recs = [r[2] for r in read_records(path)]
    if assume_simplitigs:
        offs, o = [], 0
        for r in recs:
            offs.append(o)
            o += len(r)
        offs.append(o)
        return compute(ctx, b"".join(recs), k, complements, want_mask, simplitig_offsets=offs)
    return compute(ctx, b"\n".join(recs), k, complements, want_mask, min_frequency=min_frequency)
