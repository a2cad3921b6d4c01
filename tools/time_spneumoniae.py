import sys, os, time, gzip
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import kmercamel_amd as kc
ctx = kc.Context(0)
recs = kc.read_records("tests/golden/spneumoniae.fa.gz")
seq = b"\n".join(r[2] if isinstance(r[2], bytes) else r[2].encode() for r in recs)
for k in (13, 23, 31, 63, 127):
    for rep in range(2):
        t = time.time()
        r = kc.compute(ctx, seq, k, True, want_mask=True)
        dt = time.time() - t
        st = r.stats
        r.close()
    print("spneumoniae k=%d: N=%d S=%d rounds=%d wall %.1f ms; device us: insert %d simplitigs %d overlap %d emit %d" % (
        k, st["n_kmers"], st["n_simplitigs"], st.get("matching_rounds", -1), dt * 1e3, st["us_insert_phase"], st["us_simplitig_phase"], st["us_overlap_phase"], st["us_emit_phase"]))
ctx.close()
