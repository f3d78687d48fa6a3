"""Probe of the host -> device path on the GPU box: pageable / pinned copy rates, host memcpy rate by thread count."""
import os, sys, time, threading
import numpy as np
import torch
print("cpus (affinity):", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
B = 1 << 28  # 1 GiB as float32 = 2^28 floats
a = np.random.default_rng(0).random(B, dtype=np.float32)
t0 = time.perf_counter(); d = torch.from_numpy(a).cuda(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"pageable H2D: {a.nbytes / dt / 1e9:.1f} GB/s")
t0 = time.perf_counter(); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"pageable H2D (2nd): {a.nbytes / dt / 1e9:.1f} GB/s")
p = torch.from_numpy(a).pin_memory()
for _ in range(2):
    t0 = time.perf_counter(); d.copy_(p, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pinned H2D: {a.nbytes / dt / 1e9:.1f} GB/s")
dst = np.empty_like(a)
for T in (1, 4, 8, 16, 32):
    def w(t):
        n = a.size; lo, hi = n * t // T, n * (t + 1) // T
        np.copyto(dst[lo:hi], a[lo:hi])
    ths = [threading.Thread(target=w, args=(t,)) for t in range(T)]
    t0 = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]; dt = time.perf_counter() - t0
    print(f"host memcpy {T:2d} threads: {a.nbytes / dt / 1e9:.1f} GB/s")
