"""Cycle stamps of a lone wave of the two-lanes-per-codeword SC kernel, one line per PR_WALK instruction with its steps decoded
(diagnostic build).  usage: python tools/pair_walk_profile.py [fixture.bin]"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import golden_io as gio
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))
import pypolar_amd
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_n1024_k512_sc_awgn.bin"
fx = gio.load_named(name)
d = pypolar_amd.PolarDecoder(fx.N, 1, fx.frozen, "float")
d.setErrorDetection(fx.crc_size, fx.crc_type)
ops, st, cyc = d.sc_profile(fx.llr[3])
leaf = {1: 'R0', 2: 'R1', 3: 'REP', 4: 'SPC', 5: 'DREP', 6: 'DSPC', 7: 'DSPC8', 8: 'TREP', 9: 'TYPE5', 10: 'REPONE8', 11: 'ZSPC8', 12: 'ZEROSPC'}
print("total cycles", int(cyc.sum()), "walks", len(ops))
print("(the decoded fields need the raw program: op byte, stage byte = leaf kind | c0 << 4)")
for i, (o, s, c) in enumerate(zip(ops, st, cyc)):
    if o == 0: continue
    print(f"{i:3d} leaf {leaf.get(int(s) & 15, '-'):8s} c0={int(s) >> 4}  cycles={int(c)}")
