"""Randomised configuration fuzz of the HIP decoders against the oracle (not part of the suite): random block length,
list size, frozen set (BB / BE / arbitrary subset), detector, systematic flag, input type (float32 / int8), memory space
(host / device pointers) and batch size, on noisy and tie-heavy rows.   usage: python tools/fuzz_vs_oracle.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, "polar-codes_amd"), os.path.join(_R, "tests")]
import oracle_binding as ob
import pypolar_amd
from pypolar_amd.construction import frozen_bits

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
n_cfg = n_rows = n_bad = n_refused = 0
kernels = {}
t_print = time.time()
while time.time() < t_end:
    if time.time() - t_print > 30:
        t_print = time.time()
        print(f"... {n_cfg} configurations, {n_rows} rows, {n_bad} mismatching", flush=True)
    logN = int(rng.integers(3, 16))
    L = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 8, 8, 8, 11, 16, 24, 32]))
    if os.environ.get("FUZZ_LANE") == "geo":  # the round-4 geometries of the lane-per-path kernel: N = 64 .. 4096, L = 2 .. 32
        logN = int(rng.choice([6, 6, 7, 7, 8, 8, 9, 10, 10, 11, 11, 12]))
        L = int(rng.choice([2, 2, 3, 4, 4, 5, 7, 8, 8, 9, 12, 16, 16, 17, 24, 32, 32]))
    elif os.environ.get("FUZZ_LANE"):  # only what the lane-per-path list kernel of round 3 takes: N = 1024, 4 < L <= 8
        logN = 10
        L = int(rng.choice([5, 6, 7, 8, 8, 8]))
    N = 1 << logN
    kind = int(rng.integers(0, 3))
    if kind == 2:
        K = int(rng.integers(1, N))
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    else:
        K = int(rng.integers(1, N))
        if N >= 32:
            K = max(8, K // 8 * 8)
        if K >= N:
            continue
        frozen = frozen_bits(N, K, float(rng.uniform(-1, 3)), "BE" if kind else "BB")
    nb = (K + 7) // 8
    dets = [0] + ([8] if nb >= 2 else []) + ([11] if nb >= 3 else []) + ([24] if nb >= 4 else []) + ([32] if nb >= 8 else [])
    det = int(rng.choice(dets)) if K % 8 == 0 else 0
    typ = "crcnr" if det in (11, 24) else "crc"
    sysm = bool(rng.integers(0, 2))
    i8 = bool(rng.integers(0, 4) == 0)
    dectype = "mixed" if (L > 1 and rng.integers(0, 5) == 0) else "float"
    if os.environ.get("FUZZ_CHAR"):  # the int8 decoders (FastSscFipChar / SclFipChar) on int8 rows (the oracle's entry takes int8-valued floats)
        dectype = "char"
        i8 = True
    B = int(rng.integers(1, 60 if N <= 2048 else (12 if N <= 8192 else 4)))
    mode = rng.integers(0, 3, B)
    llr = np.where(mode[:, None] == 0, rng.normal(0.5, 1.0, (B, N)) * 6.0,
                   np.where(mode[:, None] == 1, rng.choice([-2.0, -1.0, 1.0, 2.0], (B, N)), rng.normal(0, 4.0, (B, N))))
    if i8:
        x = np.clip(np.round(llr), -127, 127).astype(np.int8)
        ref_in = x.astype(np.float32)
    else:
        x = llr.astype(np.float32)
        ref_in = x
    try:
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, ref_in, systematic=sysm, dectype=dectype)
    except ValueError:
        try:
            pypolar_amd.PolarDecoder(N, L, frozen, dectype)
            print("MISMATCH: the product accepts a frozen set the reference refuses", N, K, L)
            n_bad += 1
        except ValueError:
            n_refused += 1
        continue
    # (small batches are routed to a sibling geometry by default: the fuzz of the throughput kernels switches that off, the general
    #  fuzz takes either side at random)
    pypolar_amd.SMALL_BATCH_ROWS = 0 if (os.environ.get("FUZZ_LANE") or rng.integers(0, 2)) else None
    try:
        d = pypolar_amd.PolarDecoder(N, L, frozen, dectype)
    except pypolar_amd.PolarAmdError as e:
        if e.code == -9:        # documented capacity limit
            continue
        raise
    if L > 1:
        kname = d.kernel_name(True)
        kernels[kname] = kernels.get(kname, 0) + 1
    d.setErrorDetection(det, typ)
    d.setSystematic(sysm)
    want_m = L > 1 and dectype == "float"
    if rng.integers(0, 2):
        import torch
        res = d.decode_batch(torch.from_numpy(x).cuda(), return_metrics=want_m)
        res = tuple(r.cpu().numpy() for r in res)
    else:
        res = d.decode_batch(x, return_metrics=want_m)
    bad = int((res[0] != e_out).any(axis=1).sum()) + int((res[1].astype(np.uint8) != e_ok).sum())
    if want_m:
        bad += int((res[2].view(np.uint32) != e_met.view(np.uint32)).any(axis=1).sum())
    n_cfg += 1
    n_rows += B
    if bad:
        n_bad += 1
        print(f"MISMATCH N={N} K={K} L={L} kind={kind} det={det} sys={sysm} i8={i8} type={dectype} B={B}: {bad} bad rows")
print(f"configurations {n_cfg}, rows {n_rows}, refused by both builders {n_refused}, mismatching configurations {n_bad}")
print("list kernels:", ", ".join(f"{k} {v}" for k, v in sorted(kernels.items())))
