#!/bin/bash
# Dynamic instruction mix of the hot kernels (one PMC pass): tools/pmc_insts.sh <outdir>
OUT=${1:-gpurun_out/pmc_insts}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY -d $R/$OUT -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > /dev/null 2>&1
cd $R
python3 - "$OUT" <<'PY'
import csv,collections,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:44]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,d in agg.items():
    if d.get('SQ_WAVE_CYCLES',0) < 2e8: continue
    w=d['SQ_WAVES']
    print(k.ljust(44),'waves %d  per wave: VALU %.0f SALU %.0f VMEM_RD %.1f VMEM_WR %.1f BR %.0f  cyc(x4) %.0f wait %.2f'%(w,d['SQ_INSTS_VALU']/w,d['SQ_INSTS_SALU']/w,d['SQ_INSTS_VMEM_RD']/w,d['SQ_INSTS_VMEM_WR']/w,d['SQ_INSTS_BRANCH']/w,d['SQ_WAVE_CYCLES']/w,d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES']))
PY
rm -f $OUT/*/*_kernel_trace.csv
