#!/bin/bash
# Round profile set for the decoder configs (run on the GPU box from the repo root): for every config
#   <tag>_<cfg>_bench.json                  bench.py JSON line incl. roofline.traffic (two rocprofv3 --pmc passes)
#   <tag>_<cfg>_rocprofv3_kernel_stats.csv  rocprofv3 --kernel-trace --stats of the same command
#   <tag>_<cfg>_rocprofv3_pmc_sq.txt        SQ instruction / wait counters per codeword (separate --pmc passes)
# under gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
TAG=${1:-r03}; shift; CFGS=${@:-cfg3 cfg2 cfg4 cfg5 cfg3_i8 cfg2_i8}
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
for CFG in $CFGS; do
  LB=18
  case $CFG in cfg5) LB=13;; cfg4) LB=17;; n1024_l16|n1024_l32|n2048_l8) LB=16;; esac
  # the kernel the decoder of this config launches, as the library reports it (pcd_kernel_name)
  KER=$(python3 $R/bench.py --config $CFG --kernel-name)
  python3 $R/bench.py --config $CFG --cpu-seconds 5 > $O/${TAG}_${CFG}_bench.json 2> $O/${CFG}_bench.err
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$CFG -o $CFG -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-pmc-traffic > /dev/null 2> $O/${CFG}_kt.err)
  cp $(ls $O/kt_$CFG/*/*kernel_stats.csv $O/kt_$CFG/*kernel_stats.csv 2>/dev/null | head -1) $O/${TAG}_${CFG}_rocprofv3_kernel_stats.csv
  rm -rf $R/gpurun_out/pmcsq_${CFG}_*
  bash $R/tools/pmc_sq.sh $CFG $KER $LB > $O/${TAG}_${CFG}_rocprofv3_pmc_sq.txt 2>&1
  python3 - "$O/${TAG}_${CFG}_bench.json" > $O/${TAG}_${CFG}_rocprofv3_pmc_hbm.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("# L2 <-> fabric traffic of the dominant kernel, two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) around bench.py children")
print("# (counters in KiB; gfx950: FETCH_SIZE tallies every 128-B request as 64 B -> bytes = 2 x reported; Infinity-Cache hits are included)")
for k in ("kernel", "kernel_ms_avg", "codewords_per_launch", "algorithmic_bytes_per_codeword", "traffic_fetch_size_raw", "traffic_write_size", "traffic", "traffic_over_algorithmic"):
    print(f"{k:34s} {r.get(k)}")
PY
  rm -rf $O/kt_$CFG
  echo "$CFG done: $(python3 -c "import json; d=json.load(open('$O/${TAG}_${CFG}_bench.json')); r=d['roofline']; print(d['value'], r['frac'], r.get('traffic_over_algorithmic'))")"
done
