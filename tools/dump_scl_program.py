import os, sys
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import golden_io as gio
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))  # trace / cycle stamps: diagnostic build (make -C polar-codes_amd diag)
import pypolar_amd
names = {32: 'F', 33: 'G', 34: 'COMBINE', 35: 'R0', 36: 'R1', 37: 'REP', 38: 'SPC', 40: 'F_S', 41: 'G_S', 42: 'C_S',
         43: 'C_W', 44: 'SPC_S', 45: 'R1_S', 46: 'G0_S', 47: 'VTOP', 48: 'WFG', 49: 'VG2', 0: 'END'}
fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, "float")
d.setErrorDetection(fx.crc_size, fx.crc_type)
ops, st, cyc = d.scl_profile(fx.llr[3])
for o, s, c in zip(ops, st, cyc):
    print(names.get(int(o), o), int(s), int(c))
