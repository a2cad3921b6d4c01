"""CLI phase times (KC_CLI_TRACE) of the 2.5 G-k-mer job for different numbers of copy threads per pinned chunk (KC_IO_THREADS):
python tools/cli_io_threads.py "4 8 16" [rounds]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
vals = (sys.argv[1] if len(sys.argv) > 1 else "4 8 16").split()
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = "/dev/shm/kc_io_%d" % os.getpid()
os.makedirs(d, exist_ok=True)
fa = os.path.join(d, "in.fa")
bench.generate(2, 0, 2500000, 1030, fasta=fa)
exe = os.path.join(ROOT, "kmercamel_amd", "bin", "kmercamel")
print("cores:", os.cpu_count(), flush=True)
for r in range(rounds):
    for v in vals:
        time.sleep(6)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "compute", "-k", "31", "-o", os.path.join(d, "out.txt"), fa], capture_output=True, text=True, env=dict(os.environ, KC_CLI_TRACE="1", KC_IO_THREADS=v))
        dt = time.perf_counter() - t0
        marks = {}
        for l in p.stderr.splitlines():
            if l.startswith("[cli] ") and ": " in l and "ms" in l:
                k, _, val = l[6:].rpartition(": ")
                try: marks[k] = float(val.split()[0])
                except ValueError: pass
        print("threads %-3s wall %.3f s | input on device %.0f | compute returned %.0f | in mapping %.0f | written %.0f" % (
            v, dt, marks.get("input on the device", -1), marks.get("kc_compute returned", -1), marks.get("result in the mapping", -1), marks.get("result written", -1)), flush=True)
import shutil
shutil.rmtree(d)
