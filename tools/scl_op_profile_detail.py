"""Per (op, stage) cycle table of the SCL decode program for one row (diagnostic kernel build, one wave on an idle
GPU).  usage: python tools/scl_op_profile_detail.py [fixture.bin]"""
import os
import sys
from collections import defaultdict

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import golden_io as gio
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))  # trace / cycle stamps: diagnostic build (make -C polar-codes_amd diag)
import pypolar_amd

names = {32: 'F', 33: 'G', 34: 'COMBINE', 35: 'R0', 36: 'R1', 37: 'REP', 38: 'SPC', 40: 'F_S', 41: 'G_S', 42: 'C_S',
         43: 'C_W', 44: 'SPC_S', 45: 'R1_S', 46: 'G0_S', 47: 'VTOP', 48: 'WFG', 49: 'VG2', 0: 'END'}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_n1024_k512_l8_crc24_awgn.bin"
fx = gio.load_named(name)
d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, "float")
d.setErrorDetection(fx.crc_size, fx.crc_type)
ops, st, cyc = d.scl_profile(fx.llr[3])
tot = cyc.sum()
agg = defaultdict(lambda: [0, 0])
for o, s, c in zip(ops, st, cyc):
    agg[(int(o), int(s))][0] += 1
    agg[(int(o), int(s))][1] += int(c)
print("total cycles", tot)
for (o, s), (n, c) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{names.get(o, o):8s} stage {s:2d}  n={n:4d} cycles={c:8d} ({100 * c / tot:5.1f}%) avg={c / n:8.0f}")
