#!/bin/bash
# per-kernel times of tools/r5_build_ab.py's variants: kernel trace of the one process, the scatter / build launches listed in order
# tools/r5_build_ab.sh records VARIANT...   (variant "" = default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r5
rm -rf $R/gpurun_out/r5/bab
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5/bab -o t -- python3 $R/tools/r5_build_ab.py "$@" 2>&1 | grep -v "^W2\|rocprofv3\|^E2" | tail -12
python3 - "$R" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/gpurun_out/r5/bab/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ("kc_bucket_scatter_wave_kernel", "kc_bucket_scatter_wave2_kernel", "kc_block_build_kernel<1, true>", "kc_block_build_kernel<1, false>", "kc_pack2bit"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if name in r["Kernel_Name"]]
    print(name, " ".join("%.2f" % x for x in d))
PY
