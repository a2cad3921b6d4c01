"""N > 1 leg of bench.py: one rank per GPU (launched by torchrun), STRONG scaling on the same synthetic record set as the
1-GPU run.  Rank r holds records [R*r/G, R*(r+1)/G) in HBM when the timed region starts; one step = one collective
kc_compute_sharded: the k-mer set is partitioned by table-block range (hash of the canonical minimizer), the ranks
exchange k-mer buckets, neighbour queries and chain-fragment pointers over RCCL (grouped ncclSend/ncclRecv inside the
library; torch.distributed only hands over the RCCL id and does the barrier / MAX reduction of the timing), and emit ONE
masked superstring, each rank keeping its slice.  value = N_kmers * steps / max-over-ranks time."""
import json
import os
import sys
import time


def shard_traffic(k):
    """HBM bytes per launch of the probe kernel on ONE rank's share, from this round's PMC profile of that share (null when absent)."""
    import bench
    path = os.path.join(bench.ROOT, "profiles", bench.TRAFFIC_ROUND + "shard_one_traffic.json")
    try:
        for kn, v in json.load(open(path)).items():
            if "kc_probe_block_kernel" in kn:
                return int(v["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def main(args, world, rank, local_rank, read_len):
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    import kmercamel_amd as kc

    k = args.k
    # KC_BENCH_BACKEND=gloo: ranks may share GPUs and exchange through the host -- for checking the path on a box with
    # fewer GPUs than ranks; the measured configuration is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("KC_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        red_dev = torch.device("cuda", local_rank)
    else:
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
        red_dev = torch.device("cpu")
    dev = torch.device("cuda", local_rank)
    ctx = kc.Context(local_rank)
    transport = "RCCL grouped ncclSend/ncclRecv (xGMI)"
    if backend == "nccl":
        # the library's own RCCL communicator; should it not come up on this node (all ranks decide together), fall back to the
        # host-staged transport over a gloo group rather than lose the measurement -- the line says which one ran
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            comm = kc.Comm.rccl(ctx, dist)
        except Exception as e:  # noqa: BLE001
            print("bench_sharded: RCCL communicator failed on rank %d: %r" % (rank, e), file=sys.stderr, flush=True)
            comm = None
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if comm is not None:
                comm.close()
            comm = kc.Comm.torch_host(ctx, dist, group=dist.new_group(backend="gloo"))
            transport = "host-staged gloo (the RCCL communicator did not come up)"
    else:
        comm = kc.Comm.torch_host(ctx, dist)
        transport = "host-staged " + backend
    lo, hi = bench.record_range(args.records, rank, world)
    host = bench.generate(args.seed, lo, hi, read_len)
    d_seq = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize()
    nbytes = int(d_seq.numel())
    del host

    def step(keep=False):
        r = kc.compute_sharded(ctx, comm, None, k, True, device_ptr=d_seq.data_ptr(), length=nbytes)
        st = r.stats
        if keep:
            return st, r
        r.close()
        return st

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    c0 = comm.stats()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    c1 = comm.stats()
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    per_rank = [None] * world
    st = stats[-1]
    mine = {"n_kmers_local": st["n_kmers_rank"], "us_probe_kernel": st["us_probe_kernel"], "queries": st["queries_sent"],
            "bytes_sent_per_step": (c1["bytes_sent"] - c0["bytes_sent"]) // max(1, args.steps),
            "us_in_collectives_per_step": (c1["us_in_collectives"] - c0["us_in_collectives"]) // max(1, args.steps),
            "phases_ms": [round(st[p] / 1000.0, 3) for p in ("us_insert_phase", "us_simplitig_phase", "us_overlap_phase", "us_emit_phase", "us_total")]}
    dist.all_gather_object(per_rank, mine)
    checks = {}
    if args.check:
        # the assembled superstring (rank 0): slices tile it, ones == N (min-one mask), length bounds
        st2, r = step(keep=True)
        off, n = r.slice()
        piece = r.array().copy()
        r.close()
        ones = torch.tensor([int(np.count_nonzero(piece < 97)), n], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        slices = [None] * world
        dist.all_gather_object(slices, (off, n))
        if rank == 0:
            N, S, L = st2["n_kmers"], st2["n_simplitigs"], st2["length"]
            pos = 0
            for o, n_r in sorted(s for s in slices if s[1]):
                assert o == pos, "slices do not tile the superstring"
                pos += n_r
            assert pos == L and int(ones[1]) == L
            assert int(ones[0]) == N, "mask is not min-one: %d ones, N = %d" % (int(ones[0]), N)
            assert N + k - 1 <= L <= N + S * (k - 1)
            checks = {"slices_tile": True, "ones_eq_N": True}
            print("check ok: N=%d S=%d L=%d over %d ranks" % (N, S, L, world), file=sys.stderr)
    if rank == 0:
        N, S = st["n_kmers"], st["n_simplitigs"]
        W = 8 * (1 if k < 32 else (2 if k < 64 else 4))
        us_probe = max(p["us_probe_kernel"] for p in per_rank)
        n_loc = max(p["n_kmers_local"] for p in per_rank)
        alg_bytes = n_loc * (2 * W + 16)
        achieved = alg_bytes / (us_probe * 1e-6) / 1e9 if us_probe else 0.0
        cfg_name = {2500000: "BASELINE configs[3] (the configs[2] set, hash-sharded)", 100000: "BASELINE configs[1] set, hash-sharded"}.get(args.records, "custom size")
        line = {
            "metric": "M distinct k-mers/s, compute -k %d (device-resident FASTA sequence -> device-resident masked superstring, ONE superstring over all ranks)" % k,
            "value": round(N * args.steps / elapsed / 1e6, 3),
            "unit": "Mkmers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1000.0, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64" if k < 32 else ("u128" if k < 64 else "u256"),
            "data": "synthetic",
            "config": {"workload": "%s: synthetic FASTA %d records x %d bp in total (rank r holds records [R*r/G, R*(r+1)/G)), uniform ACGT, default_rng([%d, chunk]) per %d records, k=%d, bidirectional"
                                   % (cfg_name, args.records, read_len, args.seed, bench.CHUNK_RECORDS, k),
                       "n_kmers": N, "n_simplitigs": S, "superstring_length": st["length"],
                       "sharding": "k-mer set by table-block range (hash of the canonical minimizer); exchanges: k-mer buckets, neighbour queries + answers, fragment pointer jumping, packed bases to the emitting rank",
                       "transport": transport},
            "fragment_jump_rounds": st["matching_rounds"], "overlap_repairs": st["overlap_repairs"],
            "per_rank": per_rank,
            "roofline": {"bound": "hbm", "kernel": "kc_probe_block_kernel<W,K,SK,NT,MM,SH=true> (neighbour probe of the slowest rank's shard: the single-GPU round-1 kernel in its sharded form)",
                         "achieved": round(achieved, 2),
                         "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / bench.HBM_PEAK_GBS, 5), "traffic": shard_traffic(k),
                         "traffic_source": "profiles/%sshard_one_traffic.json (PMC passes of ONE rank's share, tools/shard_one_stats.sh; null: not collected this round)" % bench.TRAFFIC_ROUND,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(us_probe, 1)},
        }
        if checks:
            line["checks"] = checks
        print(json.dumps(line), flush=True)
    comm.close()
    ctx.close()
    dist.destroy_process_group()
    return 0
