"""world_size-2 gloo test (CPU, no GPU) of the hash-sharded path's host side: the partition function the library itself
uses (kc_shard_geometry / kc_shard_owner_of, host code of libkmercamel_amd.so), the host transport bench.py / the tests
hand to kc_comm_create_callbacks (kmercamel_amd.api.host_transport: all-to-all(v) + all-gather over torch.distributed),
driven through the same three exchanges as the GPU flow:
  1. bucket exchange: every rank bins its canonical k-mers by owner and ships the buckets; the owners' sets partition the
     global set (every k-mer on exactly one rank, the one kc_shard_owner_of names);
  2. cross-rank neighbour queries: the 8 de Bruijn neighbours of every owned k-mer are looked up at THEIR owners (keys out,
     answers back in request order); the degrees equal those of a single-process computation;
  3. the locality the design relies on: most neighbour lookups stay on the asking rank.
The HIP kernels are not run here."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json, ctypes
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch, torch.distributed as dist
import gen
import kmercamel_amd.api as api

K = 21
MASK = (1 << (2 * K)) - 1
CODE = {65: 0, 67: 1, 71: 2, 84: 3}

def rc(x):
    r = 0
    for _ in range(K):
        r = (r << 2) | (3 - (x & 3)); x >>= 2
    return r

def canon(x):
    r = rc(x)
    return x if x < r else r

def kmers_of(rec):
    out, cur, n = [], 0, 0
    for ch in rec:
        c = CODE.get(ch)
        if c is None:
            cur, n = 0, 0
            continue
        cur = ((cur << 2) | c) & MASK; n += 1
        if n >= K:
            out.append(canon(cur))
    return out

def neighbours(x):
    res = []
    for c in range(4):
        res.append(canon(((x << 2) & MASK) | c))
        res.append(canon((x >> 2) | (c << (2 * (K - 1)))))
    return res

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
a2a, ag = api.host_transport(dist)
U64P = ctypes.POINTER(ctypes.c_uint64)

def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)

def allgather_u64(v):
    inp = np.array([v], dtype=np.uint64); out = np.zeros(world, dtype=np.uint64)
    assert ag(None, ptr(inp).value, ptr(out).value, 8) == 0
    return [int(x) for x in out]

def exchange(items_by_dest, dtype):
    # items_by_dest: list (per destination rank) of numpy arrays -> list (per source rank) of numpy arrays
    send = np.concatenate([np.asarray(x, dtype=dtype) for x in items_by_dest]) if any(len(x) for x in items_by_dest) else np.zeros(0, dtype=dtype)
    item = np.dtype(dtype).itemsize
    sb = np.array([len(x) * item for x in items_by_dest], dtype=np.uint64)
    # every rank tells every rank what it sends: G all-gathers of one word (the library all-gathers G words at once)
    mat = [allgather_u64(int(b)) for b in sb]          # mat[p][q] = bytes rank q sends to rank p
    rb = np.array([mat[rank][q] for q in range(world)], dtype=np.uint64)
    recv = np.zeros(int(rb.sum()) // item, dtype=dtype)
    send = np.ascontiguousarray(send)
    assert a2a(None, ptr(send).value if send.size else 0, sb.ctypes.data_as(U64P), ptr(recv).value if recv.size else 0, rb.ctypes.data_as(U64P)) == 0
    out, o = [], 0
    for q in range(world):
        n = int(rb[q]) // item
        out.append(recv[o:o + n]); o += n
    return out

recs = gen.stress_reads(genome_len=6000, n_reads=160, read_len=90, seed=4, n_frac=0.01, lower_frac=0.0)
mine = recs[len(recs) * rank // world: len(recs) * (rank + 1) // world]
local = sorted(set(x for r in mine for x in kmers_of(r)))
occ_total = sum(allgather_u64(sum(max(0, len(r) - K + 1) for r in mine)))
geoms = [api.shard_geometry(K, occ_total, world, r) for r in range(world)]
g = geoms[rank]
assert g["nblocks"] %% world == 0 and g["blocks_per_rank"] * world == g["nblocks"]
assert all(geoms[r]["first_slot"] == r * g["slots_per_rank"] for r in range(world))      # the ranges tile the table
assert g["slots_per_rank"] == g["blocks_per_rank"] << g["bshift"]

# 1. bucket exchange
owner = api.shard_owner_of(K, occ_total, world, np.array(local, dtype=np.uint64).reshape(-1, 1))
buckets = [np.array([x for x, o in zip(local, owner) if o == p], dtype=np.uint64) for p in range(world)]
got = exchange(buckets, np.uint64)
owned = sorted(set(int(x) for part in got for x in part))
own_chk = api.shard_owner_of(K, occ_total, world, np.array(owned, dtype=np.uint64).reshape(-1, 1))
assert all(int(o) == rank for o in own_chk)                                                # every k-mer sits on its owner
owned_set = set(owned)

# 2. neighbour queries
qkeys = []
for x in owned:
    qkeys.extend(neighbours(x))
qown = api.shard_owner_of(K, occ_total, world, np.array(qkeys, dtype=np.uint64).reshape(-1, 1))
order = [[i for i, o in enumerate(qown) if o == p] for p in range(world)]
requests = exchange([np.array([qkeys[i] for i in order[p]], dtype=np.uint64) for p in range(world)], np.uint64)
answers = exchange([np.array([1 if int(x) in owned_set else 0 for x in part], dtype=np.uint8) for part in requests], np.uint8)
present = [0] * len(qkeys)
for p in range(world):
    assert len(answers[p]) == len(order[p])                                                 # answers come back in request order
    for i, a in zip(order[p], answers[p]):
        present[i] = int(a)
degree_sum = sum(present)
local_fraction = sum(1 for o in qown if o == rank) / max(1, len(qown))

# single-process reference on rank 0: the whole set, brute force
all_sizes = allgather_u64(len(owned))
all_deg = allgather_u64(degree_sum)
if rank == 0:
    full = set(x for r in recs for x in kmers_of(r))
    assert sum(all_sizes) == len(full), (all_sizes, len(full))                              # the owners' sets partition the set
    want = sum(1 for x in full for y in neighbours(x) if y in full)
    assert sum(all_deg) == want, (all_deg, want)
    print(json.dumps({"ok": True, "world": world, "kmers": len(full), "sizes": all_sizes, "degree_sum": want, "local_fraction": round(local_fraction, 3)}))
assert local_fraction > 0.75, local_fraction                                               # minimizer blocks keep neighbours together
dist.barrier()
dist.destroy_process_group()
"""


def test_gloo_world2_partition_buckets_and_neighbour_queries(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(WORKER % {"root": ROOT}))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert '"ok": true' in p.stdout


def test_bench_record_ranges_tile_the_set():
    """bench.py's strong-scaling split: the ranks' record ranges tile the set and the chunked generator gives every world
    size the same records"""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    whole = bench.generate(2, 0, 3000, 130)
    for world in (1, 2, 3, 8):
        rng = [bench.record_range(3000, r, world) for r in range(world)]
        assert rng[0][0] == 0 and rng[-1][1] == 3000 and all(rng[i][1] == rng[i + 1][0] for i in range(world - 1))
        parts = [bench.generate(2, lo, hi, 130) for lo, hi in rng]
        joined = np.concatenate([np.concatenate([p, np.array([10], dtype=np.uint8)]) for p in parts])[:-1]
        assert np.array_equal(joined, whole)
