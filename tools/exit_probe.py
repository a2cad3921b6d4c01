"""process start / exit overheads of a HIP process by what it has done (tools/exit_probe.hip)"""
import subprocess, time, os
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exit_probe.bin")
for mode, normal in ((0, 0), (1, 0), (1, 1), (2, 0), (3, 0), (4, 0), (4, 1), (1, 0)):
    res = []
    for rep in range(3):
        time.sleep(3 if mode >= 3 else 0.5)
        e0 = time.time()
        p = subprocess.run([exe, str(mode), str(normal)], capture_output=True, text=True)
        e1 = time.time()
        tm, tx = map(float, p.stdout.split())
        res.append((tm - e0, tx - tm, e1 - tx))
    print("mode %d %s: " % (mode, "exit()" if normal else "_exit") + "  ".join("start %.0f ms, main %.0f ms, after exit %.0f ms" % (a * 1e3, b * 1e3, c * 1e3) for a, b, c in res), flush=True)
