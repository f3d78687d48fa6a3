"""cw/s of the int8 list decoder for several list sizes (device-resident int8 rows, kernel time from the library's
events).  usage: python tools/char_list_rate.py [log2_batch]"""
import os
import sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), _R]
import numpy as np
import torch
import pypolar_amd
from pypolar_amd.construction import frozen_bits

lb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(5)
for N, K, L in ((1024, 512, 2), (1024, 512, 4), (1024, 512, 8), (1024, 512, 16), (1024, 512, 32), (4096, 2048, 8)):
    B = 1 << (lb if N <= 1024 else lb - 2)
    fz = np.asarray(frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
    d = pypolar_amd.PolarDecoder(N, L, fz, "char")
    d.setErrorDetection(8, "crc")
    rows = torch.from_numpy(np.clip(np.rint(rng.normal(8.0, 8.0 * 1.2, (B, N))), -128, 127).astype(np.int8)).cuda()
    d.decode_batch(rows)
    d.set_timing(True)
    for _ in range(3):
        d.decode_batch(rows)
    torch.cuda.synchronize()
    ms = np.mean(d.kernel_times_ms()[-3:])
    print(f"char N={N} K={K} L={L}: {B / ms / 1e3:.3f} M cw/s ({ms:.2f} ms for 2^{int(np.log2(B))})", flush=True)
