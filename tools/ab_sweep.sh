#!/bin/bash
# usage: ab_sweep.sh N L lib1 lib2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; N=$1; L=$2; shift; shift
for round in 1 2; do for lib in "$@"; do
v=$(PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/variants/$lib SWEEP_N=$N SWEEP_L=$L python3 $R/tools/perf_sweep.py 2>/dev/null | tail -1 | awk '{print $5, $8}')
echo "N=$N L=$L round $round $lib $v"; done; done
