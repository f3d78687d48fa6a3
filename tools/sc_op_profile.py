"""Per (op, stage) cycle table of the SC decode program for one row (diagnostic kernel build, one wave of four
codewords on an idle GPU).  usage: python tools/sc_op_profile.py [fixture.bin]"""
import os
import sys
from collections import defaultdict

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import golden_io as gio
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))  # trace / cycle stamps: diagnostic build (make -C polar-codes_amd diag)
import pypolar_amd

names = {1: 'F', 2: 'G', 3: 'G0R', 4: 'COMBINE', 5: 'COMBINE0R', 6: 'RONE', 7: 'R0', 8: 'R1', 9: 'REP', 10: 'SPC',
         11: 'DREP', 12: 'DSPC', 13: 'DSPC8', 14: 'TREP', 15: 'TYPE5', 16: 'REPONE8', 17: 'ZSPC8', 18: 'ZEROSPC', 0: 'END',
         96: 'F', 97: 'G', 98: 'G0R', 99: 'C', 100: 'C0R', 101: 'RONE', 102: 'R0', 103: 'R1', 104: 'REP', 105: 'SPC', 106: 'DREP', 107: 'DSPC',
         108: 'DSPC8', 109: 'TREP', 110: 'TYPE5', 111: 'REPONE8', 112: 'ZSPC8', 113: 'ZEROSPC'}  # (96..: sc_pair_kernel, local stages)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_n1024_k512_sc_awgn.bin"
fx = gio.load_named(name)
d = pypolar_amd.PolarDecoder(fx.N, 1, fx.frozen, "float")
d.setErrorDetection(fx.crc_size, fx.crc_type)
ops, st, cyc = d.sc_profile(fx.llr[3])
tot = int(cyc.sum())
print("total cycles", tot, "instructions", len(ops))
agg = defaultdict(lambda: [0, 0])
for o, s, c in zip(ops, st, cyc):
    agg[(int(o), int(s))][0] += 1
    agg[(int(o), int(s))][1] += int(c)
for (o, s), (n, c) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(names.get(o, o)):10s} stage {s:2d}  n={n:3d} cycles={c:7d} ({100 * c / tot:5.1f}%) avg={c / n:7.0f}")
byop = defaultdict(lambda: [0, 0])
for (o, s), (n, c) in agg.items():
    byop[o][0] += n
    byop[o][1] += c
print()
for o, (n, c) in sorted(byop.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(names.get(o, o)):10s} n={n:3d} cycles={c:7d} ({100 * c / tot:5.1f}%) avg={c / n:7.0f}")
