#!/bin/bash
# A/B of library builds over several bench configs on ONE box: bash tools/ab_many.sh "cfg5 n2048_l8 ..." libA.so libB.so   (paths relative to polar-codes_amd/lib/variants)
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFGS=$1; shift
for CFG in $CFGS; do for round in 1 2; do for L in "$@"; do
    v=$(PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/variants/$L python3 $R/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-pmc-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3))")
    echo "$CFG round $round $L $v M cw/s"
done; done; done
