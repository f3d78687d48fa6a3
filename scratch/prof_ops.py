import sys, os
sys.path[:0] = ['polar-codes_amd', 'tests']
import numpy as np, golden_io as gio, pypolar_amd
names = {32:'F',33:'G',34:'COMBINE',35:'R0',36:'R1',37:'REP',38:'SPC',0:'END'}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_n1024_k512_l8_crc24_awgn.bin"
fx = gio.load_named(name)
d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, "float"); d.setErrorDetection(fx.crc_size, fx.crc_type)
ops, st, cyc = d.scl_profile(fx.llr[3])
tot = cyc.sum()
print("total cycles", tot, "instructions", len(ops))
for o in sorted(set(ops)):
    m = ops == o
    print(f"{names.get(o,o):8s} n={m.sum():4d} cycles={cyc[m].sum():8d} ({100*cyc[m].sum()/tot:5.1f}%) avg={cyc[m].mean():8.0f}")
for o in (32,33,35,36,38):
    for s in sorted(set(st[ops==o])):
        m = (ops==o)&(st==s)
        print(f"   {names[o]:8s} stage {s:2d}: n={m.sum():3d} avg={cyc[m].mean():8.0f}")
