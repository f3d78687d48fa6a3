import sys, os
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np, golden_io as gio, oracle_binding as ob, pypolar_amd
rng = np.random.default_rng(1)
for N in (16, 32, 64, 128, 256, 512):
    fxs = gio.load(os.path.join(gio.GOLDEN_DIR, f"sweep_n{N}.bin"))
    fsets = {}
    for f in fxs: fsets[f.frozen.tobytes()] = f.frozen
    for frozen in fsets.values():
        K = N - frozen.size
        for L in (2, 4, 8, 16, 32):
            B = 256
            llr = (rng.choice([-1.0, 1.0], (B, N)) * rng.choice([1.0, 2.0, 3.0], (B, N))).astype(np.float32)
            d = pypolar_amd.PolarDecoder(N, L, frozen, "float"); d.setErrorDetection(0, "crc")
            out, ok, met = d.decode_batch(llr, return_metrics=True)
            e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 0, llr, True, "float")
            nb_out = int((out != e_out).any(axis=1).sum()); nb_met = int((met != e_met).any(axis=1).sum())
            if nb_out or nb_met:
                r = int(np.argwhere((met != e_met).any(axis=1) | (out != e_out).any(axis=1))[0][0])
                print(f"N={N} K={K} L={L}: bad out rows {nb_out} met rows {nb_met}; first row {r}")
                np.save(f"/tmp/fail_N{N}_K{K}_L{L}.npy", llr[r])
                np.save(f"/tmp/fail_N{N}_K{K}_L{L}_frozen.npy", frozen)
print("done")
