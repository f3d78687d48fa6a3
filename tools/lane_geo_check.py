#!/usr/bin/env python3
"""The lane-per-path kernel under its other geometries against the oracle: bytes, verdicts, metrics bit for bit.
usage: lane_geo_check.py [N:K:L[:kind[:dectype]] ...]   (default: a grid over the geometries of csrc/scl_lane_geo.h)"""
import os, sys
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np
import oracle_binding as ob
import pypolar_amd

DEFAULT = ["512:256:8", "2048:1024:8", "1024:512:16", "1024:512:32", "4096:2048:32", "4096:2048:20", "512:128:6", "512:400:8", "2048:512:7",
           "2048:1500:8", "1024:300:12", "1024:800:24", "4096:1024:32", "4096:3000:17", "512:256:5:rand", "2048:1024:8:rand"]
bad = 0
for spec in (sys.argv[1:] or DEFAULT):
    f = spec.split(":")
    N, K, L = int(f[0]), int(f[1]), int(f[2])
    kind = f[3] if len(f) > 3 else "BB"
    dectype = f[4] if len(f) > 4 else "float"
    rng = np.random.default_rng(N + K + L)
    if kind == "rand":
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    else:
        frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, kind), dtype=np.uint32)
    det = 8
    rows = 24 if N <= 2048 else 10
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))).astype(np.float32) * np.where(rng.random((rows, N)) < 0.5, 1, -1).astype(np.float32)
    noise = rng.standard_normal((rows // 3, N)).astype(np.float32)
    ties = rng.integers(-3, 4, (rows // 2, N)).astype(np.float32)
    llr = np.concatenate([awgn, noise, ties])
    if dectype == "char":
        llr = np.clip(np.round(llr * 8), -128, 127).astype(np.float32)
    for systematic in (True, False):
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr, systematic, dectype)
        d = pypolar_amd.PolarDecoder(N, L, frozen, dectype)
        d.setErrorDetection(det, "crc")
        d.setSystematic(systematic)
        x = llr.astype(np.int8) if dectype == "char" else llr
        out, ok, met = d.decode_batch(x, return_metrics=True)
        o_ok = np.array_equal(out, e_out)
        v_ok = np.array_equal(ok.astype(np.uint8), e_ok)
        m_ok = np.array_equal(met.view(np.uint32), e_met.view(np.uint32))
        wrong_rows = np.nonzero((out != e_out).any(axis=1))[0][:8].tolist()
        print(f"{spec:24s} sys={int(systematic)} {d.kernel_name(True):28s} bytes {'ok' if o_ok else 'DIFF'} verdicts {'ok' if v_ok else 'DIFF'} metrics {'ok' if m_ok else 'DIFF'}"
              + ("" if o_ok else f" rows {wrong_rows}") + ("" if m_ok else f" metric rows {np.nonzero((met.view(np.uint32) != e_met.view(np.uint32)).any(axis=1))[0][:8].tolist()}"), flush=True)
        bad += 0 if (o_ok and v_ok and m_ok) else 1
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
