"""Deterministic synthetic inputs shared by tests and bench.py (own generator, numpy default_rng)."""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_records(n_records, length, seed):
    """n_records x length uniform i.i.d. ACGT, as a list of bytes."""
    rng = np.random.default_rng(seed)
    a = _ACGT[rng.integers(0, 4, size=(n_records, length), dtype=np.uint8)]
    return [a[i].tobytes() for i in range(n_records)]


def random_joined(n_records, length, seed):
    """the same records joined by '\\n' into one uint8 array (what kc_compute takes)"""
    rng = np.random.default_rng(seed)
    a = np.empty((n_records, length + 1), dtype=np.uint8)
    a[:, :length] = _ACGT[rng.integers(0, 4, size=(n_records, length), dtype=np.uint8)]
    a[:, length] = 10
    return a.reshape(-1)[:-1]


def stress_reads(genome_len=20000, n_reads=400, read_len=150, seed=7, n_frac=0.01, lower_frac=0.1, dup_frac=0.1):
    """Parity stress set: random genome, reads from both strands, some N, some lower case, duplicated reads."""
    rng = np.random.default_rng(seed)
    genome = _ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    reads = []
    for _ in range(n_reads):
        p = int(rng.integers(0, genome_len - read_len))
        r = genome[p:p + read_len].copy()
        if rng.random() < 0.5:
            r = np.array([comp[int(c)] for c in r[::-1]], dtype=np.uint8)
        m = rng.random(read_len)
        r[m < n_frac] = ord("N")
        low = (rng.random(read_len) < lower_frac) & (r != ord("N"))
        r[low] = r[low] + 32
        reads.append(r.tobytes())
    for _ in range(int(n_reads * dup_frac)):
        reads.append(reads[int(rng.integers(0, n_reads))])
    return reads
