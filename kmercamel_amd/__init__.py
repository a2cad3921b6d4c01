"""kmercamel_amd: MI355X-native BIGREEDY hot path of KmerCamel (k-mer set -> simplitigs -> greedy overlap path ->
masked superstring) as hand-written HIP kernels for gfx950 behind a C ABI (include/kmercamel_amd.h).

The package holds only what the path needs: csrc/ (HIP kernels + C ABI), src/ (C++ drop-in headers + CLI) and
api.py (ctypes mirror of the reference's operator interface, used by tests and bench.py)."""
from . import api  # noqa: F401
from .api import (Context, KmerSet, Simplitigs, KcError, ReadKMers, ReadKMersFiltered, get_simplitigs, simplitigs_from_strings,  # noqa: F401
                  simplitigs_from_fasta, OverlapHamiltonianPath, SuperstringFromPath, Global, LowerBoundLength, Optimize,
                  compute, compute_file, read_records, Comm, compute_sharded, compute_sharded_simplitigs)

__version__ = "0.1.0"
