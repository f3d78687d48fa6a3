#!/bin/bash
# Cost of the parts of the two-lanes-per-codeword SC kernel under full load: bench time with a part switched off (diagnostic
# build, results are wrong then).  4 = no epilogue, 8 = no leaves, 16 = no way up (COMBINEs), 32 = no way down (f / g);
# of the way down only: 64 = no pass over the channel row, 128 = no stage 8 -> 7, 256 = no stage 7 -> 6, 512 = no register stages
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg2}
export PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/libpolar_amd_diag.so
for ab in ${ABL:-0 4 8 16 32 60 64 128 256 512}; do
  ms=$(PCD_ABLATE=$ab python3 $R/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-pmc-traffic 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "ablate=$ab ms_per_step=$ms"
done
