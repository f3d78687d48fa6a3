"""Randomised configuration fuzz of the HIP encoder and puncturer against the oracle (not part of the suite).
usage: python tools/fuzz_encoder_vs_oracle.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, "polar-codes_amd"), os.path.join(_R, "tests")]
import oracle_binding as ob
import pypolar_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cfg = n_rows = n_bad = 0
t_print = time.time()
while time.time() < t_end:
    if time.time() - t_print > 30:
        t_print = time.time()
        print(f"... {n_cfg} configurations, {n_rows} rows, {n_bad} mismatching", flush=True)
    N = 1 << int(rng.integers(3, 16))
    K = int(rng.integers(1, N))
    frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    nb = (K + 7) // 8
    dets = [0] + ([8] if nb >= 2 else []) + ([11] if nb >= 3 else []) + ([24] if nb >= 4 else []) + ([32] if nb >= 8 else [])
    det = int(rng.choice(dets))
    sysm = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 50 if N <= 4096 else 6))
    info = rng.integers(0, 256, (B, nb), dtype=np.uint8)
    if K % 8:
        info[:, -1] &= (0xFF << (8 - K % 8)) & 0xFF
    e = pypolar_amd.PolarEncoder(N, frozen)
    e.setErrorDetection(det, "crcnr" if det in (11, 24) else "crc")
    e.setSystematic(sysm)
    mine = info.copy()
    if rng.integers(0, 2):
        import torch
        t = torch.from_numpy(mine).cuda()
        cw = e.encode_batch(t).cpu().numpy()
        mine = t.cpu().numpy()
    else:
        cw = e.encode_batch(mine)
    bad = 0
    for r in range(B):
        i2, c2 = ob.encode(N, frozen, det, info[r], sysm)
        bad += int(not (np.array_equal(cw[r], c2) and np.array_equal(mine[r], i2)))
    # puncturer on the same frozen set
    M = int(rng.integers(max(N // 2 + 1, N - frozen.size), N + 1))
    try:
        pos, parent = ob.punc_positions(M, frozen)
        p = pypolar_amd.Puncturer(M, frozen)
        x = rng.normal(0, 3, (B, N)).astype(np.float32)
        pu = p.puncture_batch(x)
        de = p.depuncture_batch(pu)
        for r in range(min(B, 4)):
            bad += int(not np.array_equal(pu[r], ob.puncture_f32(pos, x[r])))
            bad += int(not np.array_equal(de[r], ob.depuncture_f32(pos, N, pu[r])))
        if M % 8 == 0:
            bad += int(not np.array_equal(p.puncture_packed_batch(cw), np.stack([ob.puncture_packed(pos, cw[r]) for r in range(B)])))
    except IndexError:
        try:
            pypolar_amd.Puncturer(M, frozen)
            bad += 1
        except IndexError:
            pass
    n_cfg += 1
    n_rows += B
    if bad:
        n_bad += 1
        print(f"MISMATCH N={N} K={K} det={det} sys={sysm} B={B} M={M}: {bad}")
print(f"configurations {n_cfg}, rows {n_rows}, mismatching configurations {n_bad}")
