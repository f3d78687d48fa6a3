import os, sys, time
t0 = time.perf_counter()
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd')]
import numpy as np
import pypolar_amd
from pypolar_amd.construction import frozen_bits
t1 = time.perf_counter()
d = pypolar_amd.PolarDecoder(1024, 8, frozen_bits(1024, 512, 0.0, "BB"), "float")
t2 = time.perf_counter()
x = np.random.default_rng(1).standard_normal((64, 1024)).astype(np.float32)
out, ok = d.decode_batch(x)
t3 = time.perf_counter()
out, ok = d.decode_batch(x)
t4 = time.perf_counter()
d2 = pypolar_amd.PolarDecoder(256, 4, frozen_bits(256, 128, 0.0, "BB"), "float")
out, ok = d2.decode_batch(x[:, :256].copy())
t5 = time.perf_counter()
print(f"import {t1 - t0:.2f} s, create {t2 - t1:.3f} s, first decode {t3 - t2:.3f} s, second decode {(t4 - t3) * 1e3:.2f} ms, another geometry create + first decode {t5 - t4:.3f} s")
