#!/bin/bash
# Where the lane kernel's branch instructions (and cycles) are: rocprofv3 --pmc on the diagnostic build with parts run twice or switched off
# (PCD_ABLATE: 16384 plain; twice: 512 F chains, 2048 COMBINE chains, 4096 leaf scans, 8192 G, 32768 the op dispatch, 1024 top-of-tree;
#  off: 4 fork tails, 64 epilogue).  bash tools/pmc_lane_ablate.sh [cfg] [kernel substring]
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg3}; KER=${2:-scl_lane_kernel}
for ab in 16384 512 2048 4096 8192 32768 1024 4 64; do
  rm -rf $R/gpurun_out/pmcla_$ab
  cd /tmp && PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/libpolar_amd_diag.so PCD_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcla_$ab -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-pmc-traffic --log2-batch 15 > /dev/null 2>&1
done
cd $R && python3 - "$KER" <<'PY'
import csv, glob, collections, sys
base = None
for ab in (16384, 512, 2048, 4096, 8192, 32768, 1024, 4, 64):
    f = glob.glob(f"gpurun_out/pmcla_{ab}/*/*counter_collection.csv")
    if not f: print(ab, "missing"); continue
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if sys.argv[1] in r['Kernel_Name']: agg[r['Counter_Name']] += float(r['Counter_Value'])
    v = {k: v / 32768 for k, v in agg.items()}
    if base is None: base = v
    print(f"ablate {ab:6d}: " + "  ".join(f"{k[3:]} {v[k]:9.1f} ({v[k] - base[k]:+8.1f})" for k in sorted(v)))
PY
