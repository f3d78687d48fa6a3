"""Cycle stamps of one wave of the lane-per-path list kernel on an otherwise idle GPU (diagnostic build): cycles per
program instruction, summed per (op, stage)."""
import os, sys
from collections import defaultdict
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))
import numpy as np
import golden_io as gio
import pypolar_amd
os.environ.setdefault('PCD_ABLATE', '256')
names = {73: "WALK", 64: 'VT', 65: 'F', 66: 'G', 67: 'G0', 68: 'C', 69: 'R0', 70: 'R1', 71: 'REP', 72: 'SPC', 0: 'END'}
if len(sys.argv) > 1 and ":" in sys.argv[1]:  # N:K:L -- a BB code with CRC-8 and AWGN rows instead of a fixture
    from types import SimpleNamespace
    _N, _K, _L = (int(v) for v in sys.argv[1].split(":"))
    _rng = np.random.default_rng(1)
    _fz = np.asarray(pypolar_amd.frozen_bits(_N, _K, 0.0, "BB"), dtype=np.uint32)
    _llr = ((1.0 + 0.7 * _rng.standard_normal((16, _N))) * np.where(_rng.random((16, _N)) < 0.5, 1, -1)).astype(np.float32)
    fx = SimpleNamespace(N=_N, L=_L, frozen=_fz, crc_size=8, crc_type="crc", llr=_llr)
else:
    fx = gio.load_named(sys.argv[1] if len(sys.argv) > 1 else "cfg3_n1024_k512_l8_crc24_awgn.bin")
dectype = os.environ.get('LANE_PROFILE_TYPE', 'float')  # char: the int8 decoder (PCD_FORCE_LANE=1), the row is rounded to int8 values (x 8)
if dectype == 'char':
    os.environ['PCD_FORCE_LANE'] = '1'
d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, dectype)
d.setErrorDetection(fx.crc_size, fx.crc_type)
os.environ['PCD_SUBPROFILE'] = '/tmp/lane_sub.txt'
nrows = int(os.environ.get('PCD_PROFILE_ROWS', '1'))  # (8: a full wave of different codewords)
row = np.ascontiguousarray(fx.llr[3:3 + nrows]).reshape(-1)
if len(sys.argv) > 2:  # second argument q: the row rounded to multiples of 1/q (equal magnitudes and metrics: the exact-tie paths)
    row = (np.round(row * float(sys.argv[2])) / float(sys.argv[2])).astype(np.float32)
if dectype == 'char':
    row = np.clip(np.rint(row * 8.0), -128, 127).astype(np.float32)
ops, st, cyc = d.scl_profile(row)
agg = defaultdict(lambda: [0, 0])
for o, s, c in zip(ops, st, cyc):
    k = (names.get(int(o), str(int(o))), int(s))
    agg[k][0] += 1
    agg[k][1] += int(c)
tot = sum(v[1] for v in agg.values())
print("total cycles in ops", tot, "instructions", len(ops))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:4s} {k[1]:2d}  n={v[0]:3d}  cycles={v[1]:8d}  avg={v[1] / v[0]:8.0f}  {100.0 * v[1] / tot:5.1f}%")

# sub-stamps: leaves: 0 = scan done, 1 = weak search + candidates, 2 = merge + tie check, 3 = seats + state; walk: 1 = combines, 2 = G
sub = defaultdict(lambda: [0, 0, 0, 0, 0, 0])
for line in open('/tmp/lane_sub.txt'):
    o, s_, c, a, b, c2, d2 = (int(x) for x in line.split())
    k = names.get(o, str(o))
    if d2 == 0:  # no forking leaf behind this dispatch (needs PCD_ABLATE=256 for the phase stamps)
        seg = [c, 0, 0, 0, 0]
        k += ' (no fork)'
    else:
        seg = [a, b - a, c2 - b, d2 - c2, c - d2]
    sub[k][5] += 1
    for i in range(5):
        sub[k][i] += seg[i]
print("segments (cycles summed over the pass; PCD_ABLATE=256): walk + leaf scan | weak search + candidates | merge + ties | seats + state | bits")
for k, v in sub.items():
    print(f"{k:5s} n={v[5]:3d} ", " ".join(f"{x:8d}" for x in v[:5]))
if os.environ.get('LANE_PROFILE_HIST'):  # distribution of the fourth segment (LN_STAMP_SELECT builds: near-tie fix-up + tie tiers of select_fork) over the forks
    seg3 = sorted(int(l.split()[6]) - int(l.split()[5]) for l in open('/tmp/lane_sub.txt') if int(l.split()[6]) != 0)
    n = len(seg3)
    print("forks", n, "fourth segment: min", seg3[0], "median", seg3[n // 2], "p75", seg3[3 * n // 4], "p90", seg3[9 * n // 10], "max", seg3[-1])
    print("  below 300 cycles:", sum(1 for v in seg3 if v < 300), " 300-2000:", sum(1 for v in seg3 if 300 <= v < 2000), " 2000-6000:", sum(1 for v in seg3 if 2000 <= v < 6000), " above:", sum(1 for v in seg3 if v >= 6000))
    print("  ", seg3)
if os.environ.get('LANE_PROFILE_VT'):  # LN_STAMP_VT builds: the four stamps of a top-of-tree pass are sums over its steps: issue of the loads | wait for the rows | levels | sink
    acc = [0, 0, 0, 0, 0, 0]
    for l in open('/tmp/lane_sub.txt'):
        o, s_, c, a, b, c2, d2 = (int(x) for x in l.split())
        if names.get(o) == 'VT':
            for i, v in enumerate((a, b, c2, d2)):
                acc[i] += v
            acc[4] += c
            acc[5] += 1
    print("VT passes", acc[5], "cycles", acc[4], ": load issue", acc[0], "| wait for the rows", acc[1], "| levels", acc[2], "| sink", acc[3], "| outside the steps", acc[4] - sum(acc[:4]))

