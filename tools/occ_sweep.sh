# throughput against resident waves per CU (diagnostic build): usage: bash tools/occ_sweep.sh cfg3 "4 8 12 16" | cfg2 "6 10 14 18"
export PYPOLAR_AMD_LIB=$PWD/polar-codes_amd/lib/libpolar_amd_diag.so
cfg=${1:-cfg3}
for b in ${2:-4 8 12 16}; do
PCD_BLOCKS_PER_CU=$b PCD_SC_BLOCKS_PER_CU=$b python3 bench.py --config $cfg --steps 5 --no-cpu-baseline --no-pmc-traffic 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg waves/CU $b', d['value'], d['roofline']['kernel_ms_avg'])"
done
