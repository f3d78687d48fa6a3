"""The two-lanes-per-codeword SC kernels (float and int8, N = 1024) against the oracle on random, tie-heavy and extreme rows over
many rates and constructions (run on the GPU box; the oracle is the checker).  Every row: bytes and verdict must be equal.
usage: [PAIR_N=512] python tools/pair_stress_vs_oracle.py [rows_per_case]"""
import os
import sys

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np
import oracle_binding as ob
import pypolar_amd
from pypolar_amd.construction import frozen_bits

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(4711)
N = int(os.environ.get("PAIR_N", "1024"))   # 512: sc_pair_n9_kernel
bad = cases = onpair = 0
for dectype in ("float", "char"):
    for K in list(range(40 * N // 1024 + 8, N - 20, 24 * N // 1024)) + [N // 2]:
        for dsnr, kind in ((0.0, "BB"), (2.0, "BB"), (-1.5, "BB"), (1.0, "BE")):
            try:
                frozen = np.asarray(frozen_bits(N, K, dsnr, kind), dtype=np.uint32)
            except Exception:
                continue
            systematic = bool(rng.integers(0, 2))
            crc = (0, 8, 32)[int(rng.integers(0, 3))]
            if crc >= K:
                crc = 0
            try:
                d = pypolar_amd.PolarDecoder(N, 1, frozen, dectype)
            except Exception:
                continue  # (a frozen set the reference's Fast-SSC refuses)
            d.setErrorDetection(crc, "crc")
            d.setSystematic(systematic)
            onpair += d.launch_info(False)["codewords_per_workgroup"] == 32
            if dectype == "char":
                amp = (1, 3, 12, 127)[int(rng.integers(0, 4))]
                llr = rng.integers(-amp, amp + 1, (B, N)).astype(np.int8)
                if amp == 127:
                    llr[rng.random((B, N)) < 0.03] = -128
                llr[B // 2:] = np.clip(np.round(8.0 * (1.0 + 0.9 * rng.standard_normal((B - B // 2, N)))), -128, 127).astype(np.int8)
                x = llr.astype(np.float32)
            else:
                llr = (1.0 + 0.9 * rng.standard_normal((B, N))).astype(np.float32)
                llr[: B // 4] = rng.integers(-3, 4, (B // 4, N)).astype(np.float32)
                llr[B // 4: B // 2] = (rng.integers(-5, 6, (B // 4, N)) / 4.0).astype(np.float32)
                llr[llr == 0] = np.where(rng.random(int((llr == 0).sum())) < 0.5, 0.0, -0.0)
                x = llr
            out, ok = d.decode_batch(llr)
            e_out, e_ok, _ = ob.decode_batch(N, 1, frozen, crc, x, systematic, dectype)
            nb = int(((out != e_out).any(axis=1) | (np.asarray(ok, bool) != np.asarray(e_ok, bool))).sum())
            cases += 1
            if nb:
                bad += 1
                print(f"{dectype} K={K} dsnr={dsnr} {kind} sys={systematic} crc={crc}: {nb} of {B} rows differ", flush=True)
    print(f"{dectype} done ({cases} cases, {onpair} on the pair kernel, {bad} bad)", flush=True)
print("pair stress:", cases, "cases,", onpair, "on the pair kernel,", bad, "with differences")
sys.exit(1 if bad else 0)
