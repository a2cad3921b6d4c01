"""world_size-2 gloo test (CPU) of the multi-GPU plumbing bench.py uses: record sharding by rank, barrier, MAX over ranks
of the elapsed time, SUM of the per-rank k-mer counts.  The HIP path itself is not run here (no GPU)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gloo_world2_aggregation(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import torch, torch.distributed as dist
        import bench, gen
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        a = gen.random_joined(50, 100, seed=bench.shard_seed(rank))
        b = gen.random_joined(50, 100, seed=bench.shard_seed(1 - rank))
        assert (a != b).any()                       # ranks work on different record shards
        elapsed, total = bench.aggregate(dist, torch.device("cpu"), 1.0 + rank, 1000.0 * (rank + 1))
        assert elapsed == 2.0 and total == 3000.0   # MAX over ranks, SUM over ranks
        dist.barrier()
        if rank == 0:
            print(json.dumps({"ok": True, "world": world}))
        dist.destroy_process_group()
    """ % (ROOT, ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert '"ok": true' in p.stdout
