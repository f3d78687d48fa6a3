#!/bin/bash
# (ablate: 512 = every F chain twice, 1024 = every top-of-tree pass twice, 2048 = every COMBINE chain twice, 4096 = every leaf scan
#  twice -- all idempotent, results stay right, the time difference to the plain run (16384) is the part's cost; 64 = no epilogue)
# Cost of the parts of the lane-per-path list kernel under full load: bench time with a part switched off (diagnostic build,
# results are wrong then).  bash tools/lane_ablate.sh [cfg]
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg3}
export PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/libpolar_amd_diag.so
for ab in 16384 65536 131072 4; do
  ms=$(PCD_ABLATE=$ab python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-pmc-traffic --log2-batch 18 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "ablate=$ab ms_per_step(2^18)=$ms"
done
