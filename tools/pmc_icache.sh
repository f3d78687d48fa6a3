#!/bin/bash
# Instruction-cache counters of the dominant kernel (run on the GPU box from the repo root):  bash tools/pmc_icache.sh cfg3 scl_decode_kernel 16
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg3}; KER=${2:-scl_decode_kernel}; LB=${3:-16}
cd /tmp && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcic_$CFG -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-pmc-traffic --log2-batch $LB > /dev/null 2>&1
cd $R && python3 - "$CFG" "$KER" "$LB" <<'PY'
import csv, glob, collections, sys
cfg, ker, lb = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/pmcic_{cfg}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ker in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
n = 1 << lb
for k, v in sorted(agg.items()): print(f"{cfg} {k:24s} total={v:.6g} per_cw={v / n:.1f}")
PY
