export PYPOLAR_AMD_LIB=$PWD/polar-codes_amd/lib/libpolar_amd_diag.so
run() { env "$@" PCD_VERBOSE=1 python3 bench.py --config cfg3 --steps 5 --no-cpu-baseline --no-pmc-traffic 2>gpurun_out/w5.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['roofline']['kernel_ms_avg'], d['payload_match_fraction'])"; grep -m1 "pcd. scl" gpurun_out/w5.err; }
run PCD_X=0
run PCD_NO_VTOP=1
run PCD_S0=6
run PCD_S0=6 PCD_BLOCKS_PER_CU=16
run PCD_S0=6 PCD_BLOCKS_PER_CU=20
run PCD_S0=6 PCD_BLOCKS_PER_CU=24
