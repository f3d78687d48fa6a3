#!/bin/bash
# lane-per-path kernel against the path-group kernels on the same box (diagnostic build: PCD_NO_LANE=1 switches the lane kernel off)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/libpolar_amd_diag.so
for cfg in cfg3 cfg4; do
  for nl in PCD_FORCE_LANE PCD_NO_LANE; do
    v=$(env $nl=1 python3 $R/bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-pmc-traffic 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel'])")
    echo "$cfg $nl=1 $v"
  done
done
