"""int8 ("char") list decoder against the oracle on tie-heavy int8 rows (run on the GPU box; the oracle is the checker).
Every row: bytes, verdict and all L metrics must be equal.  usage: python tools/char_stress_vs_oracle.py [rows_per_case]"""
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np
import oracle_binding as ob
import pypolar_amd
from pypolar_amd.construction import frozen_bits

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(20260)
bad = cases = 0
for N in (64, 256, 1024, 2048):
    for rate in (0.25, 0.5, 0.75):
        K = int(N * rate) // 8 * 8
        frozen = np.asarray(frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
        for L in ((2, 4, 8, 16, 32) if N <= 1024 else (2, 4, 8)):
            for crc in ((0, "crc"), (8, "crc")):
                for amp in (1, 3, 12, 127):   # magnitudes 0 ... amp: the smaller, the more equal metrics
                    llr = rng.integers(-amp, amp + 1, (B, N)).astype(np.int8)
                    if amp == 127:
                        llr[rng.random((B, N)) < 0.02] = -128
                    d = pypolar_amd.PolarDecoder(N, L, frozen, "char")
                    d.setErrorDetection(*crc)
                    out, ok, met = d.decode_batch(llr, return_metrics=True)
                    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, crc[0], llr.astype(np.float32), True, "char")
                    nb = int(((out != e_out).any(axis=1) | (np.asarray(ok, bool) != np.asarray(e_ok, bool)) | (met != e_met).any(axis=1)).sum())
                    cases += 1
                    if nb:
                        bad += 1
                        print(f"N={N} K={K} L={L} crc={crc[0]} amp={amp}: {nb} of {B} rows differ", flush=True)
    print(f"N={N} done ({cases} cases, {bad} bad)", flush=True)
print("char stress:", cases, "cases,", bad, "with differences")
sys.exit(1 if bad else 0)
