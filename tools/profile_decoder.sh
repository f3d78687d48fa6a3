#!/bin/bash
# Kernel-trace summary of a decoder bench config (run on the GPU box from the repo root):
#   bash tools/profile_decoder.sh cfg3   ->  gpurun_out/prof_<cfg>_kernel_stats.csv, gpurun_out/b_<cfg>_prof.json
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; CFG=${1:-cfg3}; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$CFG -o $CFG -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-pmc-traffic > $O/b_${CFG}_prof.json 2> $O/b_${CFG}_prof.err &&
cp $(ls $O/prof_$CFG/*/*kernel_stats.csv $O/prof_$CFG/*kernel_stats.csv 2>/dev/null | head -1) $O/prof_${CFG}_kernel_stats.csv && head -3 $O/prof_${CFG}_kernel_stats.csv
