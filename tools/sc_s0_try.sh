export PYPOLAR_AMD_LIB=$PWD/polar-codes_amd/lib/libpolar_amd_diag.so
for s in 8 7; do
PCD_VERBOSE=1 PCD_SC_S0=$s python3 bench.py --config cfg2 --steps 5 --no-cpu-baseline --no-pmc-traffic 2>gpurun_out/s0_$s.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('s0 $s', d['value'], d['roofline']['kernel_ms_avg'], d['payload_match_fraction'])"
grep -m1 "pcd. sc" gpurun_out/s0_$s.err
done
