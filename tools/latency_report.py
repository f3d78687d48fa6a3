"""Per-call latency of the reference's actual interface: decode_vector is ONE codeword per call (decoder_python.cc:41-74,
main_benchmark.cpp:229); the reference takes 2.78 us (SC) / 254 us (SCL L = 8) per call on its CPU.  This
tool times pcd_decode_f32 from HOST pointers (numpy in, numpy out: H2D copy + kernel + D2H copy + synchronisation) and
from device pointers for B = 1, 64, 4096 rows -- latency per call and the rate it amounts to.
usage: python tools/latency_report.py [out.txt]"""
import os
import sys
import time

_R = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np
import torch
import golden_io as gio
import pypolar_amd

lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


say("# per-call latency, host pointers (numpy -> numpy) and device pointers (torch tensors), median of the timed calls")
say(f"# {'decoder':34s} {'rows':>6s} {'host us/call':>13s} {'host cw/s':>12s} {'device us/call':>15s} {'device cw/s':>12s}")
for name, label in [("cfg2_n1024_k512_sc_awgn.bin", "SC N=1024 K=512 (FastSscAvxFloat)"),
                    ("cfg3_n1024_k512_l8_crc24_awgn.bin", "SCL L=8 N=1024 K=512 CRC-24C"),
                    ("mix_n1024_k512_l8_crc24_awgn.bin", "mixed (SC, list on failures) L=8")]:
    fx = gio.load_named(name)
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype)
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    for B in (1, 64, 4096):
        x = np.ascontiguousarray(np.tile(fx.llr, ((B + fx.count - 1) // fx.count, 1))[:B])
        xd = torch.from_numpy(x).cuda()
        for _ in range(5):
            d.decode_batch(x)
            d.decode_batch(xd)
        torch.cuda.synchronize()
        reps = 200 if B <= 64 else 50
        th, td = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            d.decode_batch(x)
            th.append(time.perf_counter() - t0)
        for _ in range(reps):
            t0 = time.perf_counter()
            d.decode_batch(xd)
            torch.cuda.synchronize()
            td.append(time.perf_counter() - t0)
        h, dv = float(np.median(th)), float(np.median(td))
        say(f"  {label:34s} {B:6d} {h * 1e6:13.1f} {B / h:12.3e} {dv * 1e6:15.1f} {B / dv:12.3e}")
say("# reference CPU (BASELINE.md, one call = one codeword): SC 2.78 us, SCL L=8 254 us")
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
