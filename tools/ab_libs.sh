#!/bin/bash
# A/B of library builds on ONE box: bash tools/ab_libs.sh cfg3 libA.so libB.so ...   (paths relative to polar-codes_amd/lib/variants)
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=$1; shift
for round in 1 2 3; do
  for L in "$@"; do
    v=$(PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/variants/$L python3 $R/bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --no-pmc-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3))")
    echo "round $round $L $v M cw/s"
  done
done
