"""Throughput of the "mixed" decoder (AdaptiveFloat: SC for every row, SCL L = 8 only where the CRC rejected the SC
result) on the cfg 3 workload, inputs resident in HBM.   usage: python tools/mixed_rate.py [ebn0_db ...]"""
import os
import sys
import time

import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "polar-codes_amd")]
import bench  # noqa: E402
import pypolar_amd  # noqa: E402

dev = torch.device("cuda", 0)
for ebn0 in [float(a) for a in sys.argv[1:]] or [1.0, 2.0, 3.0]:
    cfg = dict(bench.CONFIGS["cfg3"], ebn0=ebn0)
    frozen = bench.config_frozen_set(cfg)
    B = 1 << 20
    llr, info = bench.make_batch(cfg, frozen, B, dev, seed=9)
    res = {}
    for typ in ("mixed", "float"):
        d = pypolar_amd.PolarDecoder(cfg["N"], cfg["L"], frozen, typ)
        d.setErrorDetection(cfg["crc"], cfg["crc_type"])
        out, ok = d.decode_batch(llr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out, ok = d.decode_batch(llr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        res[typ] = (B / dt, float(ok.float().mean()), float((out == info).all(dim=1).float().mean()))
    print(f"Eb/N0 {ebn0:.1f} dB: mixed {res['mixed'][0] / 1e6:.1f} M cw/s (ok {res['mixed'][1]:.4f}, payload match "
          f"{res['mixed'][2]:.4f}) | list only {res['float'][0] / 1e6:.1f} M cw/s (ok {res['float'][1]:.4f}, match {res['float'][2]:.4f})")
    del llr, info
