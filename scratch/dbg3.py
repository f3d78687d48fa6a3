import sys, os, glob
sys.path[:0] = ['polar-codes_amd', 'tests']
import numpy as np, oracle_binding as ob, pypolar_amd
# failing rows captured by dbg2 (copied into scratch/)
for f in sorted(glob.glob("scratch/fail_*_frozen.npy")):
    frozen = np.load(f); llr = np.load(f.replace("_frozen", ""))
    tag = os.path.basename(f)
    N = llr.size; L = int(tag.split("_L")[1].split("_")[0])
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float"); d.setErrorDetection(0, "crc")
    tg = d.scl_trace(llr)
    to = ob.scl_trace(N, L, frozen, llr)
    n = to.shape[0]
    bad = [i for i in range(n) if not np.array_equal(tg[i], to[i])]
    print(tag, "leaves", n, "first bad leaf", bad[:1])
    if bad:
        i = bad[0]
        # which op is leaf i?  print program leaf kinds via oracle tree? just print neighbours
        print(" prev gpu", tg[i-1]); print(" prev ora", to[i-1])
        print(" gpu ", tg[i]); print(" ora ", to[i])
