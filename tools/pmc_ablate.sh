#!/bin/bash
# Instruction-count attribution of the SCL kernel: rocprofv3 --pmc (no trace) on the diagnostic build with parts of
# the work switched off (PCD_ABLATE bits: 1 select, 2 weak search, 8 wide F/G, 16 small F/G/COMBINE bodies,
# 32 extraction, 64 all leaves, 128 every op; 256 = nothing off).  Run on the GPU box from the repo root.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
for ab in 256 257 258 264 272 288 320 384; do
  cd /tmp && PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/libpolar_amd_diag.so PCD_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/pmca_$ab -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc-traffic --log2-batch 15 > /dev/null 2>&1
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
