import sys, os
sys.path[:0] = ['polar-codes_amd', 'tests']
import numpy as np, golden_io as gio, oracle_binding as ob, pypolar_amd
N, L, det = 512, 32, 32
rng = np.random.default_rng(N * 131 + L * 7 + det)
cands = gio.load(os.path.join(gio.GOLDEN_DIR, "sweep_n512.bin"))
frozen = cands[len(cands)//2].frozen
K = N - frozen.size
B = 48
rows = []
for i in range(B):
    mode = i % 3
    if mode == 0: rows.append(rng.normal(0, 1, N))
    elif mode == 1:
        info = rng.integers(0, 256, (K + 7) // 8, dtype=np.uint8)
        _, cw = ob.encode(N, frozen, det, info)
        rows.append(1.0 - 2.0 * np.unpackbits(cw)[:N] + rng.normal(0, 0.9, N))
    else: rows.append(rng.choice([-1.0, 1.0], N) * rng.choice([1.0, 2.0], N))
llr = np.asarray(rows, np.float32)
d = pypolar_amd.PolarDecoder(N, L, frozen, "float"); d.setErrorDetection(det, "crc")
out, ok, met = d.decode_batch(llr, return_metrics=True)
e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr, True, "float")
print("K", K, "out eq", np.array_equal(out, e_out))
bad = np.argwhere(met != e_met)
print("n bad", len(bad))
for r, p in bad[:20]:
    print(r, p, met[r, p], e_met[r, p], "row mode", r % 3)
r = bad[0][0]
print(met[r]); print(e_met[r])
