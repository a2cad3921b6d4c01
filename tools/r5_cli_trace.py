"""CLI phase trace at cfg[2] (or N records): wall, [cli]/[io] trace lines, md5 of the output.  tools/r5_cli_trace.py [records] [runs]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
records = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = "/dev/shm/kc_r5"
os.makedirs(d, exist_ok=True)
fa = d + "/in.fa"
if not os.path.exists(fa):
    bench.generate(2, 0, records, 1030, fasta=fa)
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
variants = [{}] * runs
if "--exit-variants" in sys.argv:
    variants = [{}, {"KC_CLI_NO_ZAP": "1"}, {}, {"KC_CLI_NO_ZAP": "1"}, {}, {"KC_CLI_NO_ZAP": "1"}]
for v in variants:
    time.sleep(6)
    if os.path.exists(d + "/out.txt"):
        os.unlink(d + "/out.txt")
    t0 = time.perf_counter(); e0 = time.time()
    p = subprocess.run([exe, "compute", "-k", "31", "-o", d + "/out.txt", fa], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1", **v))
    e1 = time.time()
    print("wall %.3f s %s (spawn epoch %.6f, reaped %.6f)" % (time.perf_counter() - t0, v, e0, e1))
    print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[cli]") or l.startswith("[io]")), flush=True)
h = hashlib.md5()
with open(d + "/out.txt", "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        h.update(blk)
print("md5", h.hexdigest(), os.path.getsize(d + "/out.txt"))
if "--keep" not in sys.argv:
    import shutil; shutil.rmtree(d)
