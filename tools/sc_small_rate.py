"""SC decoder rate for short blocks (device-resident rows).  usage: python tools/sc_small_rate.py"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'polar-codes_amd')]
import numpy as np, torch, bench, pypolar_amd
from pypolar_amd.construction import frozen_bits
dev = torch.device("cuda", 0)
for N in (64, 128, 256, 512, 1024, 2048):
    K = N // 2
    cfg = dict(N=N, K=K, L=1, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.0)
    fz = frozen_bits(N, K, 0.0, "BB")
    B = 1 << 20
    llr, info = bench.make_batch(cfg, fz, B, dev, seed=5)
    d = pypolar_amd.PolarDecoder(N, 1, fz, "float"); d.setErrorDetection(0, "crc")
    d.decode_batch(llr); d.set_timing(True)
    for _ in range(4): d.decode_batch(llr)
    torch.cuda.synchronize()
    ms = float(np.mean(d.kernel_times_ms()[-4:]))
    print(f"N={N}: {B / ms / 1e3:.1f} M cw/s ({ms:.3f} ms)", flush=True)
