"""Real-data shapes through the C ABI (tools/shapes.py): a genome with planted repeats (long chains, the segment-parallel walks,
cycles) and 30x coverage reads with errors (multiplicity, tips, bubbles).  At 1/100 scale: size-independent properties (ones == N,
the set read back off the output, determinism) and byte identity between the library's alternative device paths; at 1/1000 -
1/500 scale (what the oracle finishes in seconds): the whole `compute` output, superstring and max-one mask, byte for byte against the
oracle's canonical-order restatement.  The reference treats all shapes alike (simplitigs.h:142-189, global.h:43-133)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import kmercamel_amd as kc
    c = kc.Context(0)
    yield c
    c.close()


def _compute(ctx, seq, k=31):
    import kmercamel_amd as kc
    r = kc.compute(ctx, seq, k, True)
    a, st = r.array().copy(), dict(r.stats)
    r.close()
    return a, st


@pytest.mark.parametrize("shape,scale", [("genome", 0.01), ("coverage", 0.01)])
def test_shape_properties_and_alternative_paths(ctx, shape, scale, monkeypatch):
    import kmercamel_amd as kc
    import shapes
    seq, _ = shapes.make_input(shape, scale)
    k = 31
    a, st = _compute(ctx, seq)
    N, S, L = st["n_kmers"], st["n_simplitigs"], st["length"]
    assert int(np.count_nonzero(a < 97)) == N            # min-one mask: one uppercase position per distinct k-mer
    assert N + k - 1 <= L <= N + S * (k - 1)
    b = kc.KmerSet(ctx, k, True, capacity_hint=L)
    b.AddKMers(a.tobytes(), case_sensitive=True)          # the k-mer set the output represents
    assert b.size() == N
    b.close()
    if shape == "genome":
        assert S * 100 < N                                # long chains: far fewer simplitigs than k-mers
    a2, _ = _compute(ctx, seq)
    assert np.array_equal(a, a2)                          # deterministic
    # the same bytes whichever device path produces them: plain chain walks, forced segment walks, direct inserts, the node-id
    # sweep instead of the build kernel's, eager compaction and no level windows in phase 2, a marking walk of its own for the cycles
    for env in ({"KC_NOFRAG": "1"}, {"KC_CHAIN_LIMIT": "64"}, {"KC_BUCKET_MIN": "100000000000"}, {"KC_NO_BUILD_IDS": "1"},
                {"KC_OV_EAGER_SHRINK": "1", "KC_OV_NO_WINDOW": "1"}, {"KC_SEG_NOMARK": "1"}, {"KC_SEG_PLAIN": "1"},
                {"KC_SEG_PLAIN": "1", "KC_SEG_DENSE_MARKS": "1"}):
        for kk, v in env.items():
            monkeypatch.setenv(kk, v)
        a3, st3 = _compute(ctx, seq)
        for kk in env:
            monkeypatch.delenv(kk)
        assert st3["n_kmers"] == N and np.array_equal(a, a3), env


@pytest.mark.parametrize("shape,scale", [("genome", 0.001), ("coverage", 0.002)])
def test_shapes_against_the_oracle(ctx, oracle, tmp_path, shape, scale):
    """The genome shape (chains of 10^5 k-mers: segment walks over contracted runs; tandem repeats: cycles, the list kernel, the
    cycle rule) and the coverage shape (branch ports, several matching rounds, hits at every level of phase 2) against the oracle,
    `-o` and `-M` bytes."""
    import kmercamel_amd as kc
    import shapes
    if shape == "genome":   # (ten times the tandem repeats of the benchmark shape: some of them close cycles even in 2.5 Mbp)
        seq = shapes.join_records(shapes.genome_records(25, max(40000, int(1e8 * scale)), 11, repeats_per_mbp=8.0, tandems_per_mbp=40.0))
    else:
        seq, _ = shapes.make_input(shape, scale)
    fa = tmp_path / (shape + ".fa")
    fa.write_bytes(b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(bytes(seq).split(b"\n"))))
    k = 31
    r = kc.compute_file(ctx, str(fa), k, True, want_mask=True)
    out, mask = r.strings(want_mask=True)
    st_dev = dict(r.stats)
    r.close()
    rc, want, wmask, st = oracle.compute_file(str(fa), k, True, False, mode=1)
    assert rc == 0 and st_dev["n_kmers"] == st[0]
    assert out == want and mask == wmask
    if shape == "genome":
        assert st_dev["cycles_cut"] > 0      # (the planted tandem repeats)
