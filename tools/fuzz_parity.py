#!/usr/bin/env python3
"""Randomised parity run on the GPU box: `python tools/fuzz_parity.py [seconds] [seed]`.  Random inputs of different shapes
(reads from a random genome with errors / N / lower case, low-complexity strings, exact repeats, circular sequences, many
copies of one read), random k in 2..127, both strand modes, both set-build paths; every case compares the whole `compute`
output (-o and -M lines) with the CPU oracle's canonical-order restatement.  Test infrastructure (uses oracle/)."""
import os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, oracle_lib
import kmercamel_amd as kc

def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))

def make_case(rng, k):
    shape = rng.choice(["reads", "lowc", "repeat", "circle", "copies", "mixed"])
    if rng.random() < 0.02:   # now and then a case beyond 2^20 positions (default bucketed build, larger overlap buckets)
        shape = "big"
        recs = [r.decode() for r in gen.stress_reads(rng.randint(200000, 1500000), rng.randint(3000, 12000), rng.randint(max(k, 100), max(k, 100) + 300), seed=rng.randint(0, 1 << 30))]
    elif shape == "reads":
        recs = [r.decode() for r in gen.stress_reads(rng.randint(500, 30000), rng.randint(20, 400), rng.randint(max(k, 30), max(k, 30) + 300), seed=rng.randint(0, 1 << 30))]
    elif shape == "lowc":
        recs = [rand_seq(rng, rng.randint(k, 5000), rng.choice(["AC", "AT", "ACG", "A"])) for _ in range(rng.randint(1, 20))]
    elif shape == "repeat":
        u = rand_seq(rng, rng.randint(k, 3000))
        recs = [u * rng.randint(2, 5), rand_seq(rng, 500) + u[: len(u) // 2]]
    elif shape == "circle":
        c = rand_seq(rng, rng.randint(k + 1, 9000))
        recs = [c + c[: k - 1], rand_seq(rng, rng.randint(k, 2000))]
    elif shape == "copies":
        r = rand_seq(rng, k + rng.randint(1, 300))
        recs = [r] * rng.randint(100, 3000) + [rand_seq(rng, 300) for _ in range(rng.randint(0, 30))]
    else:
        recs = [rand_seq(rng, rng.randint(0, 3 * k + 50), rng.choice(["ACGT", "ACGTN", "ACGTacgt"])) for _ in range(rng.randint(1, 300))]
    return shape, recs

def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rng = random.Random(seed)
    ctx, orc = kc.Context(0), oracle_lib.Oracle()
    t0, n, shapes = time.time(), 0, {}
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "case.fa")
        while time.time() - t0 < budget:
            k = rng.choice([2, 3, 5, 8, 11, 13, 15, 16, 17, 23, 31, 32, 33, 47, 63, 64, 65, 100, 127, rng.randint(2, 127)])
            compl = rng.random() < 0.7
            os.environ["KC_BUCKET_MIN"] = rng.choice(["0", "0", "1048576"])
            if rng.random() < 0.25: os.environ["KC_CHAIN_LIMIT"] = str(rng.choice([2, 64, 700]))
            else: os.environ.pop("KC_CHAIN_LIMIT", None)
            if rng.random() < 0.3: os.environ["KC_SEG_PLAIN"] = "1"      # long chains: plain segment walks instead of hops over contracted runs
            else: os.environ.pop("KC_SEG_PLAIN", None)
            # small blocks / short local probe runs: block-local spills, second homes, bucket overflows
            # (the class passes of the chain walks are only taken from 2^27 k-mers on by default: half of the cases force them)
            if rng.random() < 0.5: os.environ["KC_WALK_CLASSES_MIN"] = "0"
            else: os.environ.pop("KC_WALK_CLASSES_MIN", None)
            if rng.random() < 0.25: os.environ["KC_OV_NO_LANE_REPLAY"] = "1"   # phase 2: small buckets from global memory, one thread each
            else: os.environ.pop("KC_OV_NO_LANE_REPLAY", None)
            if rng.random() < 0.5: os.environ["KC_STREAM_LANES"] = rng.choice(["1", "2", "8"])   # lanes per simplitig when the recordings are copied out
            else: os.environ.pop("KC_STREAM_LANES", None)
            if rng.random() < 0.3: os.environ["KC_EMIT_LANES"] = rng.choice(["8", "16", "64"])   # lanes per vertex when the superstring is spelt
            else: os.environ.pop("KC_EMIT_LANES", None)
            if rng.random() < 0.2: os.environ["KC_BSHIFT"] = str(rng.choice([9, 10, 11]))
            else: os.environ.pop("KC_BSHIFT", None)
            if rng.random() < 0.2: os.environ["KC_LOCAL_PROBES"] = str(rng.choice([2, 8, 16]))
            else: os.environ.pop("KC_LOCAL_PROBES", None)
            shape, recs = make_case(rng, k)
            with open(fa, "w") as f:
                for i, r in enumerate(recs):
                    f.write(">r%d\n%s\n" % (i, r))
            # which entry: plain compute (canonical order), -S (records are simplitigs; reference order, layout-free), -z
            entry = rng.choice(["plain", "plain", "plain", "S", "z"])
            if entry == "S":
                recs = [r for r in recs if len(r) >= k and set(r) <= set("ACGT")] or [rand_seq(rng, k + 5)]
                with open(fa, "w") as f:
                    for i, r in enumerate(recs):
                        f.write(">r%d\n%s\n" % (i, r))
                rc, want, wmask, st = orc.compute_file(fa, k, compl, True, mode=0)
            elif entry == "z":
                zf = rng.choice([2, 3])
                rc, want, wmask, st = orc.compute_file_z(fa, k, compl, mode=1, min_frequency=zf)
            else:
                rc, want, wmask, st = orc.compute_file(fa, k, compl, False, mode=1)
            if os.environ.get("KC_FUZZ_VERBOSE"):   # leave a trace (and the input) in case the device call never returns
                import shutil
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                shutil.copy(fa, os.path.join(ROOT, "gpurun_out", "fuzz_last.fa"))
                print("case %d: %s k=%d compl=%s shape=%s env=%s" % (n, entry, k, compl, shape, {v: os.environ.get(v) for v in ("KC_BUCKET_MIN", "KC_CHAIN_LIMIT", "KC_SEG_PLAIN", "KC_BSHIFT", "KC_LOCAL_PROBES")}), flush=True)
            try:
                if entry == "S": r = kc.compute_file(ctx, fa, k, compl, want_mask=True, assume_simplitigs=True)
                elif entry == "z": r = kc.compute_file(ctx, fa, k, compl, want_mask=True, min_frequency=zf)
                else: r = kc.compute_file(ctx, fa, k, compl, want_mask=True)
            except kc.KcError as e:
                if rc == 0:
                    keep = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.fa" % (seed, n))
                    os.makedirs(os.path.dirname(keep), exist_ok=True)
                    os.replace(fa, keep)
                    print("GPU failed (%s) where the oracle succeeded: %s k=%d compl=%s shape=%s seed=%d case=%d env=%s -> %s" % (
                        e, entry, k, compl, shape, seed, n, {v: os.environ.get(v) for v in ("KC_BUCKET_MIN", "KC_CHAIN_LIMIT", "KC_SEG_PLAIN", "KC_BSHIFT", "KC_LOCAL_PROBES")}, keep))
                    sys.exit(1)
                n += 1
                continue
            out, mask = r.strings(want_mask=True)
            ok = rc == 0 and out == want and mask == wmask and r.stats["n_kmers"] == st[0]
            r.close()
            if ok and entry == "plain" and rng.random() < 0.3:   # maskopt on the result, both targets
                for alg, mini in (("max-one", False), ("min-one", True)):
                    rc2, wopt = orc.maskopt(out, k, compl, mini)
                    ok = ok and rc2 == 0 and kc.Optimize(ctx, out, k, compl, alg) == wopt
            if not ok:
                keep = os.path.join(ROOT, "gpurun_out", "fuzz_fail_%d_%d.fa" % (seed, n))
                os.makedirs(os.path.dirname(keep), exist_ok=True)
                os.replace(fa, keep)
                print("MISMATCH case %d (%s): k=%d compl=%s shape=%s env=%s -> %s" % (n, entry, k, compl, shape, {e: os.environ.get(e) for e in ("KC_BUCKET_MIN", "KC_CHAIN_LIMIT", "KC_SEG_PLAIN", "KC_BSHIFT", "KC_LOCAL_PROBES")}, keep))
                sys.exit(1)
            shapes[shape] = shapes.get(shape, 0) + 1
            n += 1
    print("fuzz ok: %d cases in %.0f s, seed %d, shapes %s" % (n, time.time() - t0, seed, shapes))
    ctx.close()

if __name__ == "__main__":
    main()
