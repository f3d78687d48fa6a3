"""PCIe-inclusive rate of the host-pointer path (numpy in, numpy out; pinned ring + copy / compute overlap inside
pcd_decode_f32, synchronous) for cfg 3 -- never bench.py's `value`, reported in DESIGN.md only."""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, "polar-codes_amd")]
import pypolar_amd
from pypolar_amd.construction import frozen_bits

N, K, L, B = 1024, 512, 8, 1 << 18
d = pypolar_amd.PolarDecoder(N, L, frozen_bits(N, K, 0.0, "BB"), "float")
d.setErrorDetection(24, "crcnr")
llr = np.random.default_rng(0).normal(0.6, 0.8, (B, N)).astype(np.float32)
d.decode_batch(llr[:4096])
d.decode_batch(llr)           # first large call: allocates the pinned ring, touches the pages
for name, arr in (("pageable numpy", llr),):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        d.decode_batch(arr)
        ts.append(time.perf_counter() - t0)
    print("per call ms:", [round(t * 1e3, 1) for t in ts])
    dt = min(ts)
    print(f"{name}: {B / dt / 1e6:.2f} M cw/s  ({B * N * 4 / dt / 1e9:.1f} GB/s of LLRs over PCIe incl. the kernel)")
# int8 rows (decode_vector(const char*) / pcd_decode_i8): a quarter of the bytes over PCIe
for dectype in ("float", "char"):
    d8 = pypolar_amd.PolarDecoder(N, L, frozen_bits(N, K, 0.0, "BB"), dectype)
    d8.setErrorDetection(24, "crcnr")
    B8 = 1 << 19
    rows = np.random.default_rng(1).integers(-30, 56, (B8, N), dtype=np.int8)  # (any rows: the rate does not depend on them)
    d8.decode_batch(rows[:4096])
    d8.decode_batch(rows)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        d8.decode_batch(rows)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"int8 rows, pageable numpy, \"{dectype}\" decoder: {B8 / dt / 1e6:.2f} M cw/s  ({B8 * N / dt / 1e9:.1f} GB/s of LLRs)")
    try:  # the same rows resident in HBM: what the kernel alone makes of them (the ceiling of the line above)
        import torch
        t8 = torch.from_numpy(rows).cuda()
        d8.decode_batch(t8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            d8.decode_batch(t8)
        torch.cuda.synchronize()
        print(f"    (the same rows from HBM: {3 * B8 / (time.perf_counter() - t0) / 1e6:.2f} M cw/s)")
    except Exception as e:  # noqa: BLE001
        print("    (device-resident run skipped:", e, ")")
try:
    import torch
    pin = torch.from_numpy(llr).pin_memory()
    out = torch.empty((B, K // 8), dtype=torch.uint8).pin_memory()
    a = pin.numpy()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        d.decode_batch(a)
        ts.append(time.perf_counter() - t0)
    dt = min(ts)
    print(f"pinned host memory: {B / dt / 1e6:.2f} M cw/s  ({B * N * 4 / dt / 1e9:.1f} GB/s)")
except Exception as e:  # noqa: BLE001
    print("pinned test skipped:", e)
