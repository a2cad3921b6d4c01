#!/usr/bin/env python3
"""bench.py -- `compute -k 31` (BIGREEDY) on synthetic FASTA, device-resident input -> device-resident masked superstring.

One step = one pass of the whole hot path (pack -> canonical k-mer set -> simplitigs -> greedy overlap path -> emit)
over one batch of synthetic records already resident in HBM.  N=1 runs BASELINE config[1] (100 000 records x 1 030 bp,
seed 1 = 100 M 31-mers).  N>1: one process per GPU (torchrun), every rank runs the same-size shard of independent records
(seed 1+rank) -- weak scaling over records with no data-path collective; value = all ranks' k-mers / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant kernel and
`cpu_baseline` (the reference's own code, oracle/_ref/kcref, on a bounded sample; N=1 only)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling there: 6.29 TB/s
# HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/collect_profile.sh ->
# tools/summarize_profile.py: separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH_SIZE x2 on gfx950 as calibrated on
# the pack kernel). Only valid for the default workload it was collected on.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_final_cfg2_traffic.json")


def measured_traffic(kernel_substr, default_workload):
    if not default_workload or not os.path.exists(TRAFFIC_FILE):
        return None
    try:
        t = json.load(open(TRAFFIC_FILE))
        for k, v in t.items():
            if kernel_substr in k:
                return int(v["hbm_bytes_per_launch"])
    except Exception:
        pass
    return None


def cpu_baseline(k, read_len, sample_records, seed):
    """Time the CPU path on a bounded sample of the same workload (same generator, fewer records)."""
    import gen
    recs = gen.random_records(sample_records, read_len, seed)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "sample.fa")
        with open(fa, "wb") as f:
            for i, r in enumerate(recs):
                f.write(b">r%d\n%s\n" % (i, r))
        kcref = os.path.join(ROOT, "oracle", "_ref", "kcref")
        if os.path.exists(kcref):
            t0 = time.time()
            p = subprocess.run([kcref, "compute", "-k", str(k), "-o", os.path.join(d, "out.txt"), fa], capture_output=True, text=True)
            dt = time.time() - t0
            st = json.loads(p.stderr.strip().splitlines()[-1])
            return {"value": round(st["n_kmers"] / dt / 1e6, 4), "unit": "Mkmers/s", "cores": 1, "kind": "reference",
                    "sample": "%d records x %d bp (same generator, seed %d): %d k-mers in %.1f s wall incl. FASTA read; phases read %.1f / simplitigs %.1f / path %.1f / emit %.1f s"
                              % (sample_records, read_len, seed, st["n_kmers"], dt, st["t_read_s"], st["t_simplitigs_s"], st["t_path_s"], st["t_emit_s"])}
        import oracle_lib
        orc = oracle_lib.Oracle()
        t0 = time.time()
        rc, out, _, st = orc.compute_file(fa, k, True, False, mode=0, want_mask=False)
        dt = time.time() - t0
        return {"value": round(st[0] / dt / 1e6, 4), "unit": "Mkmers/s", "cores": 1, "kind": "port",
                "sample": "%d records x %d bp (seed %d): %d k-mers in %.1f s (oracle restatement, reference order)" % (sample_records, read_len, seed, st[0], dt)}


def shard_seed(rank):
    """records of rank r are generated with seed 1+r (independent shards, weak scaling)"""
    return 1 + rank


def aggregate(dist, dev, elapsed, n_kmers):
    """(max over ranks of elapsed, sum over ranks of the k-mer counts); identity without a process group"""
    if dist is None:
        return float(elapsed), float(n_kmers)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    nk = torch.tensor([float(n_kmers)], dtype=torch.float64, device=dev)
    dist.all_reduce(nk, op=dist.ReduceOp.SUM)
    return float(t.item()), float(nk.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--records", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=0, help="default k-1+1000 (1030 for k=31)")
    ap.add_argument("--cpu-sample-records", type=int, default=40000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify size-independent properties of the output")
    args = ap.parse_args()
    k = args.k
    read_len = args.read_len or (k - 1 + 1000)

    import numpy as np
    import torch
    import gen
    import kmercamel_amd as kc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # KC_BENCH_BACKEND=gloo is for checking the N > 1 path on a box with fewer GPUs than ranks (ranks then share devices and
    # the two scalars are reduced through the host); the real thing is one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("KC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    ctx = kc.Context(local_rank)
    host = gen.random_joined(args.records, read_len, seed=shard_seed(rank))
    d_seq = torch.from_numpy(host).to(dev)
    torch.cuda.synchronize()
    nbytes = int(d_seq.numel())

    def step():
        r = kc.compute(ctx, None, k, True, want_mask=False, device_ptr=d_seq.data_ptr(), length=nbytes)
        st = r.stats
        r.close()
        return st

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st = step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed, total_kmers = aggregate(dist, red_dev, elapsed, stats[-1]["n_kmers"])

    if args.check and rank == 0:
        # size-independent properties of the result at the benchmark's own size (no oracle involved): min-one mask (ones == N),
        # length bounds, and a sampled round trip -- k-mers read off random upper-case positions of the output are in the
        # input's set, k-mers read off random input positions are in the set the output represents (same sizes).
        import ctypes
        r = kc.compute(ctx, None, k, True, want_mask=False, device_ptr=d_seq.data_ptr(), length=nbytes)
        N, S, L = r.stats["n_kmers"], r.stats["n_simplitigs"], r.length
        o = np.empty(L + 1, dtype=np.uint8)
        ctx.check(ctx.lib.kc_result_copy(r.h, o.ctypes.data_as(ctypes.c_char_p), None))
        r.close()
        o = o[:L]
        assert int(np.count_nonzero(o < 97)) == N, "mask is not min-one"
        assert N + k - 1 <= L <= N + S * (k - 1), "superstring length out of bounds"
        a = kc.KmerSet(ctx, k, True, capacity_hint=nbytes)
        a.add_device(d_seq.data_ptr(), nbytes)
        b = kc.KmerSet(ctx, k, True, capacity_hint=L)
        b.AddKMers(o.tobytes(), case_sensitive=True)
        assert a.size() == N and b.size() == N, "round trip: set sizes differ (%d, %d, %d)" % (a.size(), b.size(), N)
        rng = np.random.default_rng(7)
        code = np.full(256, 255, dtype=np.uint8)
        for i, ch in enumerate(b"ACGT"):
            code[ch] = i; code[ch + 32] = i
        def kmers_at(buf, pos):
            out = []
            for p0 in pos:
                w = code[buf[p0:p0 + k]]
                if len(w) < k or (w == 255).any():
                    continue
                f = 0; rc = 0
                for j, c in enumerate(w.tolist()):
                    f = (f << 2) | c
                    rc |= (3 - c) << (2 * j)
                out.append(min(f, rc))
            return out
        up = np.flatnonzero(o[: min(L, 50_000_000)] < 97)
        qa = kmers_at(o, rng.choice(up, size=min(2000, len(up)), replace=False))
        assert qa and all(a.contains(qa)), "round trip: an output k-mer is not in the input set"
        seq_host = d_seq[: min(nbytes, 50_000_000)].cpu().numpy()
        qb = kmers_at(seq_host, rng.integers(0, len(seq_host) - k, size=2000))
        assert qb and all(b.contains(qb)), "round trip: an input k-mer is not represented by the output"
        a.close(); b.close()
        print("check ok: N=%d S=%d L=%d, %d + %d sampled k-mers round-tripped" % (N, S, L, len(qa), len(qb)), file=sys.stderr)

    if rank == 0:
        st = stats[-1]
        W = 8 * (1 if k < 32 else (2 if k < 64 else 4))
        N, S = st["n_kmers"], st["n_simplitigs"]
        # dominant kernel: whichever of the two hash kernels took longer (both timed with HIP events recorded on the
        # library's own stream around the launch, averaged over the timed steps)
        us_probe = sum(s["us_probe_kernel"] for s in stats) / len(stats)
        us_insert = sum(s["us_insert_kernel"] for s in stats) / len(stats)
        if us_probe >= us_insert:
            name, us, ksub = "kc_choose_dense_kernel<W,K> (neighbour probe, round 1)", us_probe, "kc_choose_dense_kernel"
            alg_bytes = N * (2 * W + 16)          # one successful probe read per side (2W) + the two 8-B port links written
        else:
            name, us, ksub = "k-mer set build (kc_bucket_scatter_kernel + kc_block_build_kernel)", us_insert, "kc_bucket_scatter_kernel"
            alg_bytes = 0.25 * nbytes + N * 2 * W  # packed 2-bit read + one probe read and one key write per distinct k-mer
        achieved = alg_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        phases = {p: round(sum(s[p] for s in stats) / len(stats) / 1000.0, 3) for p in
                  ("us_insert_phase", "us_simplitig_phase", "us_overlap_phase", "us_emit_phase", "us_total")}
        line = {
            "metric": "M distinct k-mers/s end-to-end compute -k %d (device-resident sequence -> masked superstring)" % k,
            "value": round(total_kmers * args.steps / elapsed / 1e6, 3),
            "unit": "Mkmers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1000.0, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64" if k < 32 else ("u128" if k < 64 else "u256"),
            "data": "synthetic",
            "config": {"workload": "synthetic FASTA %d records x %d bp per GPU, uniform ACGT, seed 1+rank, k=%d, bidirectional (BASELINE configs[1] at 1 GPU)"
                                   % (args.records, read_len, k),
                       "n_kmers_per_gpu": N, "n_simplitigs": S, "superstring_length": st["length"], "sharding": "independent record shards per rank, no collective"},
            "phases_ms": phases,
            "matching_rounds": st["matching_rounds"], "overlap_repairs": st["overlap_repairs"],
            "roofline": {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(ksub, args.records == 100000 and k == 31 and not args.read_len),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_us": round(us, 1),
                         "whole_path_alg_bytes": int(N * (1.5 + 4 * W + 16) + S * 2 * k * (2 * W + 8)),
                         "whole_path_achieved_GBs": round((N * (1.5 + 4 * W + 16) + S * 2 * k * (2 * W + 8)) / (phases["us_total"] * 1e-3) / 1e9, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(k, read_len, args.cpu_sample_records, seed=1)
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
