#!/bin/bash
# Encoder profile (run on the GPU box from the repo root): kernel stats, then FETCH_SIZE / WRITE_SIZE / SQ instruction
# counters in separate --pmc passes (never combined with tracing).
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc -o enc -- python3 $R/bench.py --config enc3 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc-traffic > $O/b_enc3_prof.json 2> $O/b_enc3_prof.err &&
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  cd /tmp && rocprofv3 --pmc $c --output-format csv -d $O/pmc_enc_$n -- python3 $R/bench.py --config enc3 --steps 2 --warmup 0 --no-cpu-baseline --no-pmc-traffic > /dev/null 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
    f = glob.glob(f"gpurun_out/pmc_enc_{n}/*/*counter_collection.csv")
    if not f: print(n, "missing"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "encode_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(n, {k: v for k, v in sorted(agg.items())})
PY
