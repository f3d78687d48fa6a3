#!/bin/bash
# SQ instruction-mix counters of the dominant kernel of a bench config (run on the GPU box from the repo root):
#   bash tools/pmc_sq.sh cfg2 sc_decode_kernel 18
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg3}; KER=${2:-scl_decode_kernel}; LB=${3:-16}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmcsq_${CFG}_$n -- python3 $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-pmc-traffic --log2-batch $LB > /dev/null 2>&1
done
cd $R && python3 - "$CFG" "$KER" "$LB" <<'PY'
import csv, glob, collections, sys
cfg, ker, lb = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/pmcsq_{cfg}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if ker in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
n = 1 << lb
print(f"# {ker}, bench.py --config {cfg} --log2-batch {lb} --steps 1: totals over the dispatch, per_cw = total / {n}")
for k, v in sorted(agg.items()): print(f"{k:24s} total={v:.6g} per_cw={v / n:.1f}")
PY
