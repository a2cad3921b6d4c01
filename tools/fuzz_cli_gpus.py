#!/usr/bin/env python3
"""Randomised check of the hash-sharded path through the CLI on the GPU box: `python tools/fuzz_cli_gpus.py [seconds] [seed]`.
Random inputs (the shapes of tools/fuzz_parity.py: reads with errors / N / lower case, low-complexity strings, repeats, circular
sequences with tails, copies), k in 18..127, 2-4 rank threads sharing the GPU, both strand modes, -M and -z now and then; every case
compares the bytes of `kmercamel compute --gpus N` with those of the plain `kmercamel compute` (which the other fuzz run holds
against the oracle)."""
import os, random, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity
BIN = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")

def run(args, timeout=120, env=None):
    p = subprocess.run([BIN] + args, capture_output=True, timeout=timeout, env=env)
    return p.returncode, p.stdout, p.stderr

def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4711
    rng = random.Random(seed)
    t0 = time.time(); n = 0; shapes = {}
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "in.fa")
        while time.time() - t0 < budget:
            k = rng.choice([18, 19, 23, 31, 32, 33, 47, 63, 64, 101, 127])
            shape, recs = fuzz_parity.make_case(rng, k)
            if shape == "big" and rng.random() < 0.7:
                continue
            if shape == "circle":   # reads that run into the circle or out of it
                c = recs[0]
                for _ in range(rng.randint(0, 4)):
                    at = rng.randrange(max(1, len(c) - k))
                    tail = fuzz_parity.rand_seq(rng, rng.randint(1, 60))
                    recs.append(c[at:at + k - 1] + tail if rng.random() < 0.5 else tail + c[at:at + k - 1])
            simp = rng.random() < 0.15   # -S: the records are simplitigs (only records that can be: k or more letters of ACGT)
            if simp:
                recs = [r.upper() for r in recs if len(r) >= k and set(r.upper()) <= set("ACGT")][:20000] or [fuzz_parity.rand_seq(rng, k + 7)]
            with open(fa, "w") as f:
                for i, r in enumerate(recs):
                    f.write(">r%d\n%s\n" % (i, r))
            uni = rng.random() < 0.3
            mask = rng.random() < 0.3 and not simp
            gpus = rng.choice([2, 2, 3, 4])
            base = ["compute", "-k", str(k)] + (["-u"] if uni else []) + (["-S"] if simp else (["-z", str(rng.choice([2, 3]))] if rng.random() < 0.25 else []))
            outs = []
            # the alternative paths of the sharded flow, now and then: pointer jumping instead of waves (or waves cut short), plain
            # walks, the two-pass bucket layout, replicated phase 2, the rolling scatter
            knobs = {}
            if rng.random() < 0.3: knobs["KC_WAVE_MAX"] = rng.choice(["0", "4", "64"])
            if rng.random() < 0.15: knobs["KC_NOFRAG"] = "1"
            if rng.random() < 0.15: knobs["KC_SHARD_TWO_PASS"] = "1"
            if rng.random() < 0.5: knobs["KC_OVG_MAX"] = rng.choice(["0", "0", "40", "400", "4000"])   # phase 2 with the keys of every table level divided among the ranks (by default only while more than 2^20 vertices are live)
            if rng.random() < 0.4:   # the sampled fragment ranking forced, now and then with tiny rates / rounds (everything unreached, no hashed splitter at all)
                knobs["KC_RANK_SAMPLED"] = rng.choice(["0", "1", "1"])
                if rng.random() < 0.3: knobs["KC_SLOT_TEST_SHRINK"] = rng.choice(["2", "9", "1000"])   # slotted exchanges that overflow and repeat
                if rng.random() < 0.5: knobs["KC_RANK_C"] = rng.choice(["1", "2", "3", "64", "4096"]); knobs["KC_RANK_R"] = rng.choice(["1", "2", "5", "40"])
            if rng.random() < 0.15: knobs["KC_SCATTER_ROLL"] = "1"
            if rng.random() < 0.25: knobs["KC_BRANCH_HOST"] = "1"   # the branch subgraph settled on the host (the device matching is the default)
            for extra in ([], ["--gpus", str(gpus)]):
                o = os.path.join(d, "o%d" % len(outs)); m = os.path.join(d, "m%d" % len(outs))
                for f in (o, m):
                    if os.path.exists(f): os.remove(f)
                rc, so, se = run(base + extra + ["-o", o] + (["-M", m] if mask else []) + [fa], env=dict(os.environ, **knobs) if extra else None)
                if rc != 0 and b"contains no k-mers" not in se:
                    print("FAILED rc", rc, base + extra, knobs, se[-2000:].decode(errors="replace")); print("seed", seed, "case", n, shape, k); sys.exit(1)
                # (an input without k-mers ends both runs with the usage text and the same exit code)
                outs.append((rc, open(o, "rb").read() if rc == 0 else b"", open(m, "rb").read() if mask and rc == 0 else b""))
            if outs[0] != outs[1]:
                keep = os.path.join(ROOT, "gpurun_out", "fuzz_cli_fail.fa")
                os.makedirs(os.path.dirname(keep), exist_ok=True)
                open(keep, "w").write(open(fa).read())
                print("MISMATCH seed", seed, "case", n, shape, "k", k, "gpus", gpus, "uni", uni, "mask", mask, "knobs", knobs, "input kept in", keep); sys.exit(1)
            n += 1; shapes[shape] = shapes.get(shape, 0) + 1
    print("cli --gpus fuzz ok: %d cases in %.0f s, seed %d, shapes %s" % (n, time.time() - t0, seed, shapes))

if __name__ == "__main__":
    main()
