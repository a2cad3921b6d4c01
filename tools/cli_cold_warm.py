"""Does a fresh CLI process pay for VRAM that the previous process has just freed?  Runs the 2.5 G-k-mer CLI job back to
back and after pauses and prints the wall time and the time spent inside hipMalloc (KC_CLI_TRACE)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
records = int(sys.argv[1]) if len(sys.argv) > 1 else 2500000
d = "/dev/shm/kc_cw_%d" % os.getpid()
os.makedirs(d, exist_ok=True)
fa = os.path.join(d, "in.fa")
bench.generate(2, 0, records, 1030, fasta=fa)
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
def run(tag):
    t0 = time.perf_counter()
    p = subprocess.run([exe, "compute", "-k", "31", "-o", os.path.join(d, "out.txt"), fa], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1"))
    dt = time.perf_counter() - t0
    pool = [l for l in p.stderr.splitlines() if "device pool" in l]
    print("%-28s wall %.3f s  %s" % (tag, dt, pool[0][6:] if pool else p.stderr[-200:]), flush=True)
run("first")
run("back to back")
run("back to back")
for pause in (3, 8, 20):
    time.sleep(pause)
    run("after %d s pause" % pause)
import shutil
shutil.rmtree(d)
