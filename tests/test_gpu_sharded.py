"""Hash-sharded multi-GPU path on a real GPU: G ranks (processes) run kc_compute_sharded together and must
produce ONE masked superstring, byte-identical to the single-GPU kc_compute on the same records.  On a 1-GPU box the
ranks share the device and exchange through the host (gloo); the kernels, the partition, the bucket exchange, the
neighbour queries, the fragment ranking and the distributed emit are the production code."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gen
import kmercamel_amd as kc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(world, fasta, k, mode, out, extra_env=None, timeout=600):
    port = 29600 + (os.getpid() + world * 7) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py"), fasta, str(k), mode, out]
    env = dict(os.environ, **(extra_env or {}))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, "ranks failed:\n" + p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return np.load(out)


def write_fasta(path, recs):
    with open(path, "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n%s\n" % (i, r))


@pytest.fixture(scope="module")
def ctx():
    c = kc.Context(0)
    yield c
    c.close()


def single(ctx, fasta, k, bi):
    r = kc.compute_file(ctx, fasta, k, bi)
    a = r.array().copy()
    r.close()
    return a


def test_self_comm_equals_single_gpu(ctx, tmp_path):
    """world = 1 through the sharded code path (no transport) == kc_compute"""
    recs = gen.random_records(300, 500, seed=11)
    comm = kc.Comm.self_(ctx)
    for k, bi in ((31, True), (31, False), (21, True), (63, True), (127, True)):
        seq = b"\n".join(recs)
        want = kc.compute(ctx, seq, k, bi)
        got = kc.compute_sharded(ctx, comm, seq, k, bi)
        assert got.slice() == (0, want.length)
        assert np.array_equal(got.array(), want.array()), (k, bi)
        want.close(); got.close()
    comm.close()


def test_ranks_against_the_oracle_directly(ctx, oracle, golden_dir, tmp_path):
    """The sharded flow against the CPU oracle itself (canonical-order restatement, oracle/kc_oracle.cpp), not only against the single-GPU
    HIP path: random reads, overlapping error-laden reads with branch ports, and the first 120 kbp of S. pneumoniae (long chains, a
    cycle); 2 - 4 ranks, both strand modes, k = 23 / 31 / 63, with the sampled fragment ranking, the plain waves and pointer jumping, and
    with the table levels of phase 2 divided among the ranks."""
    import gzip
    cases = []
    fa1 = str(tmp_path / "reads.fa")
    write_fasta(fa1, gen.random_records(300, 450, seed=21))
    cases += [(fa1, 31, "bi", 3, None), (fa1, 31, "uni", 2, None), (fa1, 63, "bi", 2, {"KC_RANK_SAMPLED": "0"})]
    fa2 = str(tmp_path / "stress.fa")
    write_fasta(fa2, [r for r in gen.stress_reads(genome_len=25000, n_reads=500, read_len=180, seed=9, n_frac=0.002, lower_frac=0.05) if len(r) >= 63])
    cases += [(fa2, 23, "bi", 4, {"KC_OVG_MAX": "0"}), (fa2, 31, "bi", 3, {"KC_WAVE_MAX": "0"}), (fa2, 63, "uni", 2, None), (fa2, 31, "bi", 2, {"KC_BRANCH_HOST": "1"})]
    fa3 = str(tmp_path / "spn.fa")
    with gzip.open(os.path.join(golden_dir, "spneumoniae.fa.gz"), "rt") as f:
        text = "".join(l.strip() for l in f if not l.startswith(">"))
    with open(fa3, "w") as f:
        f.write(">spn_head\n%s\n" % text[:120000])
    cases += [(fa3, 31, "bi", 3, {"KC_RANK_SAMPLED": "1", "KC_RANK_C": "4", "KC_RANK_R": "6"}), (fa3, 23, "bi", 2, None)]
    for fa, k, mode, world, env in cases:
        rc, want, _, st = oracle.compute_file(fa, k, mode == "bi", False, mode=1, want_mask=False)
        assert rc == 0
        got = run_ranks(world, fa, k, mode, str(tmp_path / "out.npy"), extra_env=env)
        assert bytes(got).decode() == want, (os.path.basename(fa), k, mode, world, env)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_emit_one_identical_superstring_random_reads(ctx, tmp_path, world):
    recs = gen.random_records(400, 600, seed=3)
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    for k, mode in ((31, "bi"), (31, "uni")):
        got = run_ranks(world, fa, k, mode, str(tmp_path / "out.npy"))
        assert np.array_equal(got, single(ctx, fa, k, mode == "bi")), (world, k, mode)


@pytest.mark.parametrize("k", [23, 63, 127])
def test_ranks_wide_keys_and_branching_reads(ctx, tmp_path, k):
    """overlapping reads from both strands with errors: branch ports, chains that span many reads; phase 2 with the overlap
    keys divided among the ranks (KC_OVG_MAX=0: every level that builds a table of its own, as the levels with more than 2^20 live
    vertices are by default) and with every rank doing all of it (what these small inputs get by default)"""
    recs = [r for r in gen.stress_reads(genome_len=30000, n_reads=600, read_len=max(200, 2 * k), seed=k, n_frac=0.002, lower_frac=0.05) if len(r) >= k]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    want = single(ctx, fa, k, True)
    # (KC_BRANCH_HOST=1: the branch subgraph settled by the host version -- sort + hash maps -- instead of the device matching)
    for world, env in ((2, {"KC_OVG_MAX": "0"}), (3, {"KC_OVG_MAX": "300"}), (2, None), (3, {"KC_BRANCH_HOST": "1"})):
        got = run_ranks(world, fa, k, "bi", str(tmp_path / "out.npy"), extra_env=env)
        assert np.array_equal(got, want), (world, env)


@pytest.mark.parametrize("world,k,mode", [(2, 31, "bi"), (3, 31, "uni"), (4, 23, "bi"), (2, 63, "bi")])
def test_ranks_real_genome_with_cycles(ctx, golden_dir, tmp_path, world, k, mode):
    """S. pneumoniae (the reference's data/spneumoniae.fa): chains of > 10^5 k-mers, branch ports, and a few cycles -- cut at
    where their best edge to a chain end is, whether they stay on one rank or span several.  One record, so some ranks hold no
    input at all."""
    fa = os.path.join(golden_dir, "spneumoniae.fa.gz")
    got = run_ranks(world, fa, k, mode, str(tmp_path / "out.npy"), timeout=900, extra_env={"KC_OVG_MAX": "0"} if world != 3 else None)
    assert np.array_equal(got, single(ctx, fa, k, mode == "bi")), (world, k, mode)


@pytest.mark.parametrize("k", [13, 31, 63, 127])
def test_ranks_simplitigs_input(ctx, tmp_path, k):
    """-S on the sharded path (simplitigs_from_fasta, simplitigs.h:89-103): the records are the simplitigs, in record order; every rank
    packs its own, phase 2 runs on the all-gathered end k-mers, the packed words travel to the rank that emits them.  Records of k
    ... 3k+50 characters with overlaps between their ends, both strand modes, 2-4 ranks (some with no record at all)."""
    import random
    rng = random.Random(31 * k)
    recs = []
    for i in range(700):
        n = rng.choice([k, k + 1, 2 * k, 3 * k + 50])
        head = recs[rng.randrange(len(recs))][-(k - 1):] if recs and rng.random() < 0.5 else b""   # an overlap with an earlier record's end
        body = bytes(rng.choice(b"ACGT") for _ in range(max(0, n - len(head))))
        recs.append((head + body)[:max(n, k)] if len(head + body) >= k else head + body + b"A" * k)
    fa = str(tmp_path / "simp.fa")
    write_fasta(fa, recs)
    for mode in ("bi", "uni"):
        r = kc.compute_file(ctx, fa, k, mode == "bi", assume_simplitigs=True)
        want = r.array().copy()
        r.close()
        for world in (2, 3, 4):
            got = run_ranks(world, fa, k, mode + "-S", str(tmp_path / "out.npy"))
            assert np.array_equal(got, want), (mode, world)
    few = str(tmp_path / "few.fa")
    write_fasta(few, recs[:2])
    r = kc.compute_file(ctx, few, k, True, assume_simplitigs=True)
    want = r.array().copy()
    r.close()
    assert np.array_equal(run_ranks(4, few, k, "bi-S", str(tmp_path / "out.npy")), want)


def test_requests_the_sharded_flow_does_not_serve_run_on_rank_0(ctx, tmp_path):
    """k < 18 (plain hashing: no minimizer blocks to shard by), the sparse branch (S >= N / 5, main.cpp:175-181: the ranks find out
    together after phase 1), -S with -M (global.h:185-195) and -M / -z combined with them: kc_compute_sharded does not leave the
    collective for these -- the ranks' records go to rank 0, which runs kc_compute, and every rank gets its slice of the one result
    (round 6; KC_SHARD_NO_ROOT_FALLBACK=1 gives the error code KC_E_UNSUPPORTED_SHARDED back instead).  Byte-identical to one GPU,
    string and mask, 2 - 4 ranks, some of them without a record."""
    def single_both(fa, k, bi, **kw):
        r = kc.compute_file(ctx, fa, k, bi, want_mask=True, **kw)
        o, m = r.strings(want_mask=True)
        r.close()
        return np.frombuffer(o.encode(), dtype=np.uint8), np.frombuffer(m.encode(), dtype=np.uint8)
    out = str(tmp_path / "out.npy")
    # k < 18
    fa = str(tmp_path / "reads.fa")
    write_fasta(fa, gen.random_records(120, 300, seed=4))
    for k, mode, world in ((15, "bi", 3), (9, "uni", 2), (17, "bi", 4)):
        wo, wm = single_both(fa, k, mode == "bi")
        got = run_ranks(world, fa, k, mode + "-M", out)
        assert np.array_equal(got, wo) and np.array_equal(np.load(out + ".mask.npy"), wm), (k, mode, world)
    # the sparse branch: 3 k-mers per record at k = 31
    sp = str(tmp_path / "sparse.fa")
    write_fasta(sp, gen.random_records(3000, 33, seed=5))
    wo, wm = single_both(sp, 31, True)
    for world in (2, 3):
        got = run_ranks(world, sp, 31, "bi-M", out)
        assert np.array_equal(got, wo) and np.array_equal(np.load(out + ".mask.npy"), wm), world
    r = kc.compute_file(ctx, sp, 31, True)
    assert r.stats["sparse_branch"] == 1
    r.close()
    # -S with -M
    import random
    rng = random.Random(77)
    recs = [bytes(rng.choice(b"ACGT") for _ in range(rng.choice([31, 40, 95]))) for _ in range(400)]
    sf = str(tmp_path / "simp.fa")
    write_fasta(sf, recs)
    for k, mode in ((31, "bi"), (21, "uni")):
        wo, wm = single_both(sf, k, mode == "bi", assume_simplitigs=True)
        for world in (2, 4):
            got = run_ranks(world, sf, k, mode + "-S-M", out)
            assert np.array_equal(got, wo) and np.array_equal(np.load(out + ".mask.npy"), wm), (k, mode, world)
    # and with the knob the code comes back, from every rank together
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29977",
                        os.path.join(ROOT, "tests", "shard_worker.py"), fa, "15", "bi", out], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, KC_SHARD_NO_ROOT_FALLBACK="1"))
    assert p.returncode != 0 and "-9" in (p.stdout + p.stderr) and "not part of the hash-sharded path" in (p.stdout + p.stderr), (p.stdout + p.stderr)[-1500:]


@pytest.mark.parametrize("k", [23, 63])
def test_ranks_min_frequency_filter(ctx, tmp_path, k):
    """-z on the sharded path (ReadKMersFiltered, parser.h:122-141): every occurrence of a k-mer, whichever rank read it, is counted
    at the k-mer's owner; the k-mers seen fewer than z times leave the table before phase 1.  Overlapping reads from both strands
    with errors: occurrence counts of 1 ... ~10."""
    recs = [r for r in gen.stress_reads(genome_len=20000, n_reads=900, read_len=max(150, 2 * k), seed=5 + k, n_frac=0.002, lower_frac=0.05) if len(r) >= k]
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    for z in (2, 3):
        for mode in ("bi", "uni"):
            r = kc.compute_file(ctx, fa, k, mode == "bi", min_frequency=z)
            want = r.array().copy()
            r.close()
            for world in (2, 3):
                got = run_ranks(world, fa, k, "%s-z%d" % (mode, z), str(tmp_path / "out.npy"))
                assert np.array_equal(got, want), (z, mode, world)


@pytest.mark.parametrize("k,mode", [(19, "bi"), (31, "bi"), (31, "uni"), (33, "bi")])
def test_ranks_cycles_next_to_chain_ends(ctx, tmp_path, k, mode):
    """circular sequences (cycles of k-mers, short ones that stay in a block and long ones that span ranks), each with reads that
    run into the circle or out of it (edges from cycle ports to dead ports: where the cycle is opened), and isolated circles
    (opened at their smallest k-mer); pointer jumping and waves"""
    import random
    rng = random.Random(k * 7 + len(mode))
    recs = []
    for i in range(60):
        n = rng.choice([k + 1, k + 5, 3 * k, 500, 4000])
        c = "".join(rng.choice("ACGT") for _ in range(n))
        recs.append((c + c[: k - 1]).encode())
        for _ in range(rng.randint(0, 3)):
            at = rng.randrange(n)
            rot = (c + c)[at: at + k - 1]
            tail = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 40)))
            recs.append((rot + tail).encode() if rng.random() < 0.5 else (tail + rot).encode())
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    want = single(ctx, fa, k, mode == "bi")
    for world, env in ((2, None), (3, {"KC_WAVE_MAX": "0"}), (4, None)):
        got = run_ranks(world, fa, k, mode, str(tmp_path / "out.npy"), extra_env=env)
        assert np.array_equal(got, want), (world, env)


def test_ranks_cfg1_full_size(ctx, tmp_path):
    """BASELINE configs[1] (10^8 31-mers) over 4 ranks == single GPU, byte for byte"""
    sys.path.insert(0, ROOT)
    import bench
    seq = bench.generate(1, 0, 100000, 1030)
    fa = str(tmp_path / "cfg1.fa")
    bench.generate(1, 0, 100000, 1030, fasta=fa)
    want = kc.compute(ctx, seq, 31, True)
    w = want.array().copy()
    want.close()
    got = run_ranks(4, fa, 31, "bi", str(tmp_path / "out.npy"), timeout=1200)
    assert np.array_equal(got, w)
    # phase 2 in its mixed form, as a 2.5 10^9-k-mer set takes it by default: the levels through the 16-mer table whole on every rank, the
    # table levels with more than KC_OVG_MAX live vertices divided among the ranks, the rest (grouped levels) whole again
    got = run_ranks(3, fa, 31, "bi", str(tmp_path / "out.npy"), timeout=1200, extra_env={"KC_OVG_MAX": "60000"})
    assert np.array_equal(got, w)


def test_rccl_transport_single_rank(ctx, tmp_path):
    """the RCCL back end (grouped ncclSend/ncclRecv inside the library, id handed over through torch.distributed) with the one
    rank a 1-GPU box allows: librccl loads, the communicator comes up, every exchange of the flow runs through it"""
    recs = gen.random_records(300, 400, seed=8)
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py"), fa, "31", "bi", str(tmp_path / "out.npy"), "--backend", "nccl"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert np.array_equal(np.load(str(tmp_path / "out.npy")), single(ctx, fa, 31, True))


@pytest.mark.parametrize("k,mode", [(31, "bi"), (31, "uni"), (63, "bi"), (23, "bi-z2")])
def test_rccl_send_recv_code_runs_on_one_gpu(ctx, tmp_path, k, mode):
    """KC_RCCL_SELF_SENDRECV=1 routes the rank's OWN part of every all-to-all through a grouped ncclSend + ncclRecv to itself instead
    of the device copy: the dlsym'ed entry points (prototypes from <rccl/rccl.h>), the datatype constant, the skipping of zero
    counts and the stream order against the kernels before and after all run on the one GPU a test box has.  Same bytes as one GPU."""
    import json
    recs = gen.stress_reads(1500, 150, 300, seed=31) if "z" in mode else gen.random_records(400, 500, seed=9)
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    port = 29800 + (os.getpid() + k) % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "shard_worker.py"), fa, str(k), mode, str(tmp_path / "out.npy"), "--backend", "nccl"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, KC_RCCL_SELF_SENDRECV="1"))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    st = json.load(open(str(tmp_path / "out.npy") + ".stats.json"))
    assert st["self_sendrecv_calls"] >= 5, st   # (the exchanges of the flow went through ncclSend / ncclRecv)
    strand, _, z = mode.partition("-z")
    if z:
        want = kc.compute_file(ctx, fa, k, strand == "bi", min_frequency=int(z))
        w = want.array().copy(); want.close()
    else:
        w = single(ctx, fa, k, strand == "bi")
    assert np.array_equal(np.load(str(tmp_path / "out.npy")), w)


def test_a_rank_that_fails_alone_releases_its_peers(tmp_path, monkeypatch):
    """In-process thread transport (kc_thread_group, what `compute --gpus N` uses): a rank that fails on its own between two exchanges
    (forced with KC_TEST_FAIL_RANK) must not leave the others waiting for it: they come back with KC_E_TRANSPORT."""
    import ctypes
    import threading
    from kmercamel_amd import api
    lib = api.load()
    world = 3
    g = ctypes.c_void_p()
    assert lib.kc_thread_group_create(world, ctypes.byref(g)) == 0
    recs = gen.random_records(600, 300, seed=4)
    monkeypatch.setenv("KC_TEST_FAIL_RANK", "1")
    codes, errs = [None] * world, [None] * world

    def rank_main(r):
        c = kc.Context(0)
        comm = kc.Comm.threads(c, g, r, world)
        mine = b"\n".join(recs[len(recs) * r // world: len(recs) * (r + 1) // world])
        try:
            res = kc.compute_sharded(c, comm, mine, 31, True)
            res.close()
            codes[r] = 0
        except kc.KcError as e:
            codes[r], errs[r] = e.code, str(e)
        comm.close()
        c.close()

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that failed"
    assert codes[1] == api.KC_E_NO_DEVICE and "fails on its own" in errs[1], (codes, errs)
    assert codes[0] == api.KC_E_TRANSPORT and codes[2] == api.KC_E_TRANSPORT, (codes, errs)
    lib.kc_thread_group_destroy(g)


def _visible_gpus():
    try:
        import torch
    except ImportError:   # (the module must still collect where torch is absent: these tests then skip)
        return 0
    return torch.cuda.device_count()   # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_visible_gpus() < 2 or not os.environ.get("KC_TEST_MULTI_GPU"),
                    reason="opt-in (KC_TEST_MULTI_GPU=1) and needs two GPUs: the grouped ncclSend/ncclRecv path between ranks has never run on this pool's 1-GPU boxes")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_transport_between_gpus(ctx, golden_dir, tmp_path, world):
    """RCCL over xGMI with one rank per GPU (runs only where the box has the GPUs): random reads, both strand modes, and the genome
    with cycles -- byte-identical to one GPU"""
    if _visible_gpus() < world:
        pytest.skip("needs %d GPUs" % world)
    recs = gen.random_records(4000, 600, seed=13)
    fa = str(tmp_path / "in.fa")
    write_fasta(fa, recs)
    for path, k, mode in ((fa, 31, "bi"), (fa, 31, "uni"), (fa, 63, "bi"), (os.path.join(golden_dir, "spneumoniae.fa.gz"), 31, "bi")):
        port = 29950 + (os.getpid() + world) % 40
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "tests", "shard_worker.py"), path, str(k), mode, str(tmp_path / "out.npy"), "--backend", "nccl"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        assert np.array_equal(np.load(str(tmp_path / "out.npy")), single(ctx, path, k, mode == "bi")), (path, k, mode)


def test_bench_gpus2_line(tmp_path):
    """bench.py --gpus 2 (its own launcher, host-staged transport because the box has one GPU): a strong-scaling line on one record set"""
    import json
    env = dict(os.environ, KC_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--records", "20000", "--steps", "2", "--warmup", "1", "--check"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["n_kmers"] == 20000 * 1000
    assert line["checks"]["ones_eq_N"] and len(line["per_rank"]) == 2


def test_fragment_ranking_variants(ctx, golden_dir, tmp_path):
    """The two fragment rankings of the sharded flow -- waves (one message per fragment; read sets) and pointer jumping (log
    rounds; long chains, unidirectional sets) -- give the same bytes: reads with KC_WAVE_MAX=0 (pointer jumping forced), the
    genome with waves forced (thousands of rounds, and its cross-rank cycles are found as the fragments no wave reaches)."""
    recs = gen.random_records(1500, 600, seed=31)
    fa = str(tmp_path / "reads.fa")
    write_fasta(fa, recs)
    want = single(ctx, fa, 31, True)
    # (default: the sampled ranking, these chains have ~60 fragments; KC_RANK_SAMPLED=0: plain waves; C / R small: nearly every fragment
    # is unreached after R rounds and becomes a splitter of its own; C huge: no hashed splitters at all)
    for knob in ({}, {"KC_WAVE_MAX": "0"}, {"KC_RANK_SAMPLED": "0"}, {"KC_RANK_SAMPLED": "1", "KC_RANK_C": "3", "KC_RANK_R": "2"},
                 {"KC_RANK_SAMPLED": "1", "KC_RANK_C": "4096", "KC_RANK_R": "5"},
                 # slots a hundredth of what was asked for: the exchanges overflow, all ranks see it in the headers and repeat with the known maximum
                 {"KC_RANK_SAMPLED": "1", "KC_SLOT_TEST_SHRINK": "100"}, {"KC_RANK_SAMPLED": "1", "KC_RANK_C": "3", "KC_RANK_R": "4", "KC_SLOT_TEST_SHRINK": "7"}):
        assert np.array_equal(run_ranks(3, fa, 31, "bi", str(tmp_path / "o.npy"), extra_env=knob), want), knob
    g = os.path.join(golden_dir, "spneumoniae.fa.gz")
    wantg = single(ctx, g, 31, True)
    assert np.array_equal(run_ranks(2, g, 31, "bi", str(tmp_path / "g.npy"), extra_env={"KC_RANK_SAMPLED": "0", "KC_WAVE_MAX": "100000000"}, timeout=900), wantg)
    # the genome (chains of > 10^5 k-mers, cycles inside and across ranks) through the sampled ranking: splitters on cycles never
    # reach a head and are given up after log2(all splitters) rounds, the cycle rule then opens them
    for knob in ({"KC_RANK_SAMPLED": "1"}, {"KC_RANK_SAMPLED": "1", "KC_RANK_C": "2", "KC_RANK_R": "3"}):
        assert np.array_equal(run_ranks(3, g, 31, "bi", str(tmp_path / "g.npy"), extra_env=knob, timeout=900), wantg), knob


def test_exchange_volume_per_kmer(ctx, tmp_path):
    """Collectives and bytes a rank sends per k-mer it owns (kc_comm_stats), 4 and 8 ranks, 2*10^6 31-mers of random 1 kbp reads.
    Round 4: 17.5 B at 4 ranks with the fragment ranking by plain waves -- and one collective pair per fragment of the longest chain
    (341 / 371 collectives at 4 / 8 ranks, most of the flow's wall time).  Round 5: the sampled ranking (waves that stop at splitters,
    pointer jumping over the splitters, one pull per fragment; slotted exchanges without a count exchange) and phase 2 without
    collectives below 2^23 vertices: <= 80 collectives whatever the number of ranks, for ~1.7x the bytes.  The
    bounds keep both from creeping back up."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    fa = str(tmp_path / "reads.fa")
    seq = bench.generate(5, 0, 2000, 1030, fasta=fa)
    out = str(tmp_path / "out.npy")
    got = run_ranks(4, fa, 31, "bi", out)
    want = kc.compute(ctx, seq, 31, True)
    assert np.array_equal(got, want.array())
    want.close()
    st = json.load(open(out + ".stats.json"))
    print("4 ranks:", st)
    assert st["world"] == 4 and max(st["bytes_sent_per_owned_kmer"]) <= 28.0 and st["collectives"] <= 80, st
    # the same over 8 ranks sharing the GPU: the number of collectives does not grow with the ranks (the plain waves needed one round per
    # fragment of the longest chain: 295 collectives of 371 at 8 ranks), and the volume stays bounded
    got = run_ranks(8, fa, 31, "bi", out, timeout=900)
    assert np.array_equal(got, np.asarray(np.load(out)))
    st8 = json.load(open(out + ".stats.json"))
    print("8 ranks:", st8)
    w2 = kc.compute(ctx, seq, 31, True)
    assert np.array_equal(got, w2.array())
    w2.close()
    assert st8["world"] == 8 and max(st8["bytes_sent_per_owned_kmer"]) <= 38.0 and st8["collectives"] <= 80, st8
