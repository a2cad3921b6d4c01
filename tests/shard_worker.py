"""One rank of a hash-sharded run, started by tests/test_gpu_sharded.py (and usable by hand):

    python -m torch.distributed.run --nproc-per-node G tests/shard_worker.py <input> <k> <bi|uni> <out.npy> [--backend gloo|nccl]

Every rank takes its contiguous share of the input's records, the ranks run kc_compute_sharded together, rank 0 collects
the slices in rank order and stores the assembled superstring.  With the gloo backend all ranks may share one GPU
(exchanges are staged through the host): that is how the multi-rank path is exercised on a 1-GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def split_records(recs, rank, world):
    lo, hi = len(recs) * rank // world, len(recs) * (rank + 1) // world
    return recs[lo:hi]


def main():
    import torch
    import torch.distributed as dist
    import kmercamel_amd as kc
    path, k, mode, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(1, ndev)
    if backend == "nccl":
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    ctx = kc.Context(dev)
    comm = kc.Comm.rccl(ctx, dist) if backend == "nccl" else kc.Comm.torch_host(ctx, dist)
    recs = [r[2] for r in kc.read_records(path)]
    mine = b"\n".join(split_records(recs, rank, world))
    # mode: "bi" / "uni", optionally with "-zN" (minimum frequency N) or "-S" (the records are simplitigs), and "-M" (the mask too:
    # stored behind the string in <out>.mask.npy)
    want_mask = "-M" in mode
    mode = mode.replace("-M", "")
    if mode.endswith("-S"):
        r = kc.compute_sharded_simplitigs(ctx, comm, split_records(recs, rank, world), k, mode.startswith("bi"), want_mask=want_mask)
    else:
        strand, _, z = mode.partition("-z")
        r = kc.compute_sharded(ctx, comm, mine, k, strand == "bi", min_frequency=int(z) if z else 1, want_mask=want_mask)
    off, n = r.slice()
    if want_mask:
        o_s, m_s = r.strings(want_mask=True)
        piece, mpiece = np.frombuffer(o_s.encode(), dtype=np.uint8), m_s.encode()
    else:
        piece, mpiece = r.array().copy(), b""
    stats = dict(r.stats)
    r.close()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((off, n, piece.tobytes(), stats, comm.stats(), mpiece), gathered, dst=0)
    if rank == 0:
        gathered.sort(key=lambda g: (g[0], g[1]))
        pos, parts, mparts = 0, [], []
        for off_r, n_r, b, _, _, mb in gathered:
            if n_r == 0:
                continue
            assert off_r == pos, "slices do not tile the superstring: offset %d, expected %d" % (off_r, pos)
            parts.append(b)
            mparts.append(mb)
            pos += n_r
        assert pos == stats["length"], "slices cover %d of %d characters" % (pos, stats["length"])
        np.save(out_path, np.frombuffer(b"".join(parts), dtype=np.uint8))
        if want_mask:
            np.save(out_path + ".mask.npy", np.frombuffer(b"".join(mparts), dtype=np.uint8))
        per_rank = [g[4]["bytes_sent"] / max(1, g[3]["n_kmers_rank"]) for g in gathered]
        print("sharded ok: world=%d N=%d S=%d L=%d jump_rounds=%d bytes_sent(rank0)=%d max bytes sent per owned k-mer=%.2f collectives=%d" % (
            world, stats["n_kmers"], stats["n_simplitigs"], stats["length"], stats["matching_rounds"], gathered[0][4]["bytes_sent"], max(per_rank),
            gathered[0][4]["calls"]), flush=True)
        import json
        with open(out_path + ".stats.json", "w") as f:
            json.dump({"world": world, "n_kmers": stats["n_kmers"], "bytes_sent_per_owned_kmer": per_rank, "collectives": gathered[0][4]["calls"],
                       "self_sendrecv_calls": gathered[0][4].get("self_sendrecv_calls", 0)}, f)
    comm.close()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
