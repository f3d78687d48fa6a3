"""Throughput of the HIP decoders over a grid of (N, L) at rate 1/2 (BB construction, CRC-8 for L > 1), AWGN at
2 dB, inputs resident in HBM, kernel time from HIP events.   usage: python tools/perf_sweep.py   (needs a GPU)"""
import os
import sys

import numpy as np
import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "polar-codes_amd")]
import bench  # noqa: E402
import pypolar_amd  # noqa: E402
from pypolar_amd.construction import frozen_bits  # noqa: E402

dev = torch.device("cuda", 0)
print(f"{'N':>6} {'L':>3} {'rows':>8} {'kernel ms':>10} {'cw/s':>12} {'info Gbit/s':>12} {'LDS B/cw':>9}  kernel")
NS = [int(x) for x in os.environ.get("SWEEP_N", "64,128,256,512,1024,2048,4096,16384").split(",")]
LS = [int(x) for x in os.environ.get("SWEEP_L", "1,2,4,8,16,32").split(",")]
for N in NS:
    for L in LS:
        K = N // 2
        cfg = dict(N=N, K=K, L=L, dsnr=0.0, crc=8 if L > 1 else 0, crc_type="crc", ebn0=2.0)
        frozen = frozen_bits(N, K, 0.0, "BB")
        if N == 16384 and L == 32:
            continue                                  # log2 N * log2 L > 64: PCD_ERR_CAPACITY
        work = N * (1 if L == 1 else 8 * L)          # keep each launch in the 5-100 ms range, >= 4 passes of the grid
        B = int(min(1 << 20, max(1 << 14, (1 << 36) // (work * 64))))
        if os.environ.get("SWEEP_LOG2B"):            # a fixed batch instead (the lane kernels with 16 / 32 codewords per wave want >= 2^17 rows to fill the chip)
            B = 1 << int(os.environ["SWEEP_LOG2B"])
        llr, info = bench.make_batch(cfg, frozen, B, dev, seed=5)
        d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
        d.setErrorDetection(cfg["crc"], "crc")
        out = torch.empty((B, (K + 7) // 8), dtype=torch.uint8, device=dev)
        ok = torch.empty((B,), dtype=torch.uint8, device=dev)
        d.decode_batch(llr, out=out, ok=ok)
        d.set_timing(True)
        for _ in range(3):
            d.decode_batch(llr, out=out, ok=ok)
        torch.cuda.synchronize()
        ms = float(np.mean(d.kernel_times_ms()))
        li = d.launch_info(L > 1)
        print(f"{N:6d} {L:3d} {B:8d} {ms:10.3f} {B / ms * 1e3:12.4g} {B / ms * 1e3 * K / 1e9:12.3f} "
              f"{li['lds_bytes_per_codeword']:9d}  {d.kernel_name(L > 1)}", flush=True)
        del llr, info, out, ok, d
