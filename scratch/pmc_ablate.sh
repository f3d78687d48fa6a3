#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for ab in 256 257 258 264 272 288 320 384; do
  cd /tmp && PCD_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmca_$ab -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --log2-batch 15 > /dev/null 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for ab in (256,257,258,264,272,288,320,384):
    f=glob.glob(f"gpurun_out/pmca_{ab}/*/*counter_collection.csv")
    if not f: print(ab,"missing"); continue
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if 'scl_decode' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(ab, {k: round(v/32768) for k,v in sorted(agg.items())})
PY
