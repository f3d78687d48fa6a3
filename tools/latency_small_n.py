"""One SC codeword per call at the small block lengths: the pair kernels (a lone wave of up to 32 codewords) against the small-batch sibling
(sc_decode_kernel, one codeword per wave).  usage: [LAT_L=8] python tools/latency_small_n.py   (needs a GPU)"""
import os, sys, time
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np, torch, pypolar_amd
from pypolar_amd.construction import frozen_bits
rng = np.random.default_rng(3)
LL = int(os.environ.get("LAT_L", "1"))   # list size (1: the SC decoders)
for N in (128, 256, 512, 1024):
    fz = frozen_bits(N, N // 2, 0.0, "BB")
    for small in (None, 0):
        pypolar_amd.SMALL_BATCH_ROWS = small
        d = pypolar_amd.PolarDecoder(N, LL, fz, "float")
        if LL > 1: d.setErrorDetection(8, "crc")
        res = []
        for B in (1, 32, 256, 2048):
            x = torch.from_numpy((1.0 + 0.8 * rng.standard_normal((B, N))).astype(np.float32)).cuda()
            for _ in range(5): d.decode_batch(x)
            torch.cuda.synchronize(); td = []
            for _ in range(200):
                t0 = time.perf_counter(); d.decode_batch(x); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
            res.append(f"B={B}: {np.median(td) * 1e6:.0f}")
        print(f"N={N} small_batch_rows={'default' if small is None else small} ({d.small_batch_rows()}):", " ".join(res), "us/call", flush=True)
