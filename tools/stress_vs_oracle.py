"""Randomised GPU-vs-oracle stress (not part of the test suite): many rows per configuration, several noise
models, bit-exact outputs + verdicts + metrics.   usage: python tools/stress_vs_oracle.py [seed] [scale]"""
import os
import sys
import time

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np
import oracle_binding as ob
import pypolar_amd
from pypolar_amd.construction import frozen_bits

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
SCALE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CASES = [(1024, 512, 8, 24, "BB", 6000), (1024, 512, 1, 0, "BB", 20000), (1024, 768, 8, 8, "BB", 3000),
         (1024, 128, 8, 11, "BB", 3000), (512, 256, 4, 8, "BE", 6000), (256, 128, 32, 8, "BB", 1500),
         (2048, 1024, 8, 24, "BB", 1500), (128, 64, 16, 8, "BE", 4000), (64, 32, 2, 8, "BB", 8000),
         (4096, 2048, 32, 32, "BB", 60),
         # list sizes above 8: rank-based selection with several candidates per lane, bit blocks in the arena
         (1024, 512, 16, 24, "BB", 1500), (1024, 512, 32, 24, "BB", 800), (512, 384, 32, 8, "BE", 800),
         (2048, 1024, 32, 32, "BB", 200), (4096, 1024, 16, 24, "BB", 150), (256, 64, 12, 8, "BB", 1500),
         (8192, 4096, 16, 32, "BB", 24)]
tot = bad = 0
for N, K, L, det, ctor, B in [(a, b, c, dd, e, f * SCALE) for a, b, c, dd, e, f in CASES]:
    F = frozen_bits(N, K, 0.0, ctor)
    nb = K // 8
    t0 = time.time()
    infos = rng.integers(0, 256, (B, nb), dtype=np.uint8)
    llr = np.empty((B, N), np.float32)
    for r in range(B):
        _, cw = ob.encode(N, F, det, infos[r])
        x = 1.0 - 2.0 * np.unpackbits(cw)[:N]
        m = r % 4
        sig = rng.uniform(0.5, 1.3)
        if m == 0:
            llr[r] = x + sig * rng.normal(0, 1, N)
        elif m == 1:
            llr[r] = np.round((x + sig * rng.normal(0, 1, N)) * 3)        # coarse grid: many ties
        elif m == 2:
            llr[r] = rng.normal(0, 1, N)
        else:
            llr[r] = np.where(rng.random(N) < 0.1, -x, x) * rng.choice([1.0, 1.0, 2.0], N)
    d = pypolar_amd.PolarDecoder(N, L, F, "float")
    d.setErrorDetection(det, "crcnr" if det in (11, 24) else "crc")
    out, ok, met = d.decode_batch(llr, return_metrics=True) if L > 1 else (*d.decode_batch(llr), None)
    e_out, e_ok, e_met = ob.decode_batch(N, L, F, det, llr)
    b_out = int((out != e_out).any(axis=1).sum())
    b_ok = int((ok.astype(np.uint8) != e_ok).sum())
    b_met = int((met.view(np.uint32) != e_met.view(np.uint32)).any(axis=1).sum()) if L > 1 else 0
    tot += B
    bad += b_out + b_ok + b_met
    print(f"N={N} K={K} L={L} det={det} {ctor}: rows={B} bad_out={b_out} bad_ok={b_ok} "
          f"metrics_not_bit_identical={b_met} ok_rate={e_ok.mean():.3f} ({time.time()-t0:.0f}s)", flush=True)
print("TOTAL rows", tot, "mismatches", bad)
sys.exit(1 if bad else 0)
