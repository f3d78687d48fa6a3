"""Throughput of the list decoder on QUANTISED LLRs (multiples of a step, int8 range): equal magnitudes and equal path
metrics are then common, and the exact-tie paths (literal emulation of the reference's selection sorts) are taken
much more often than on real-valued input.   usage: python tools/quantized_input_rate.py"""
import os
import sys

import numpy as np
import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "polar-codes_amd")]
import bench  # noqa: E402
import pypolar_amd  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["cfg3"])
frozen = bench.config_frozen_set(cfg)
B = 1 << 19
llr, info = bench.make_batch(cfg, frozen, B, dev, seed=3)
d = pypolar_amd.PolarDecoder(cfg["N"], cfg["L"], frozen, "float")
d.setErrorDetection(cfg["crc"], cfg["crc_type"])
out = torch.empty((B, 64), dtype=torch.uint8, device=dev)
ok = torch.empty((B,), dtype=torch.uint8, device=dev)
for name, x in (("float32 as generated", llr),
                ("rounded to 1/8 (6-bit-like)", torch.clamp(torch.round(llr * 8.0), -127, 127) / 8.0),
                ("int8 (rounded x 8)", torch.clamp(torch.round(llr * 8.0), -127, 127).to(torch.int8)),
                ("rounded to 1/2 (3-bit-like)", torch.clamp(torch.round(llr * 2.0), -7, 7) / 2.0)):
    d.decode_batch(x, out=out, ok=ok)
    d.set_timing(True)
    for _ in range(3):
        d.decode_batch(x, out=out, ok=ok)
    torch.cuda.synchronize()
    ms = float(np.mean(d.kernel_times_ms()))
    d.set_timing(False)
    print(f"{name:32s}: {B / ms * 1e3 / 1e6:6.2f} M cw/s, detector pass {float(ok.float().mean()):.4f}, "
          f"payload match {float((out == info).all(dim=1).float().mean()):.4f}")
