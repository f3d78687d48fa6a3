export PYPOLAR_AMD_LIB=$PWD/polar-codes_amd/lib/libpolar_amd_diag.so
for cfgv in "PCD_S0=7" "PCD_S0=7 PCD_SB=8" "PCD_S0=6 PCD_BLOCKS_PER_CU=10" "PCD_S0=6 PCD_BLOCKS_PER_CU=8"; do
  echo "== $cfgv"; env $cfgv PCD_VERBOSE=1 python3 bench.py --config cfg5 --steps 3 --no-cpu-baseline --no-pmc-traffic 2> /tmp/e.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['payload_match_fraction'])"; grep "\[pcd\] scl" /tmp/e.txt | head -1
done
