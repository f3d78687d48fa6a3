"""Debug aid: lane-per-path kernel vs the oracle on one (K, L, construction) case; per-leaf trace comparison of the first bad row."""
import os, sys
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))
import numpy as np
import oracle_binding as ob
import pypolar_amd
N = 1024
for K, L, kind in [(512, 5, "rand"), (512, 8, "rand"), (512, 5, "BB"), (300, 7, "rand")]:
    rng = np.random.default_rng(N + K + (5 if kind == "rand" and K == 512 else L))
    frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32) if kind == "rand" else np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, kind), dtype=np.uint32)
    llr = (1.0 + 0.8 * rng.standard_normal((40, N))).astype(np.float32) * np.where(rng.random((40, N)) < 0.5, 1, -1).astype(np.float32)
    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, llr, True, "float")
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    d.setErrorDetection(8, "crc")
    out, ok, met = d.decode_batch(llr, return_metrics=True)
    bad = np.where((out != e_out).any(axis=1) | (met.view(np.uint32) != e_met.view(np.uint32)).any(axis=1))[0]
    print(K, L, kind, "bad rows", len(bad), flush=True)
    if len(bad):
        r = bad[0]
        tr = d.scl_trace(llr[r]); et = ob.scl_trace(N, L, frozen, llr[r])
        n = et.shape[0]
        for i in range(n):
            if not np.array_equal(tr[i].view(np.uint32), et[i].view(np.uint32)):
                print("  first differing leaf", i, "of", n)
                for k in range(max(0, i - 2), min(n, i + 2)):
                    print("   got", k, tr[k]); print("   ref", k, et[k])
                break
        else:
            print("  traces equal over", n, "leaves; metrics", met[r], e_met[r])
