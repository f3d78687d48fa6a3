"""Lane-per-path kernels against the path groups at the small block lengths over the rates (diagnostic build: PCD_FORCE_LANE / PCD_NO_LANE).
usage: THR_N=256|512 [THR_K=16,32,...] [THR_TYPE=char] python tools/lane_threshold_small_n.py   (needs a GPU)"""
import os, sys
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [_R, os.path.join(_R, 'polar-codes_amd')]
import numpy as np, torch, bench, pypolar_amd
from pypolar_amd.construction import frozen_bits
dev = torch.device("cuda", 0)
TYPE = os.environ.get("THR_TYPE", "float")
N = int(os.environ.get("THR_N", "256"))
KS = [int(k) for k in os.environ.get("THR_K", "16,32,48,64,200,230,245").split(",")]
for L in [int(l) for l in os.environ.get("THR_L", "8,2").split(",")]:
    for K in KS:
        fz = frozen_bits(N, K, 0.0, "BB")
        cfg = dict(N=N, K=K, L=L, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0)
        B = 1 << 18
        llr, info = bench.make_batch(cfg, fz, B, dev, seed=5)
        if TYPE == "char":
            llr = torch.clamp(torch.round(llr * 8), -128, 127).to(torch.int8)
        res = []
        for mode in ("PCD_NO_LANE", "PCD_FORCE_LANE"):
            os.environ.pop("PCD_NO_LANE", None); os.environ.pop("PCD_FORCE_LANE", None)
            os.environ[mode] = "1"
            d = pypolar_amd.PolarDecoder(N, L, fz, TYPE); d.setErrorDetection(8, "crc")
            out = torch.empty((B, (K + 7) // 8), dtype=torch.uint8, device=dev); ok = torch.empty((B,), dtype=torch.uint8, device=dev)
            d.decode_batch(llr, out=out, ok=ok); d.set_timing(True)
            for _ in range(3): d.decode_batch(llr, out=out, ok=ok)
            torch.cuda.synchronize()
            res.append((d.kernel_name(True), B / float(np.mean(d.kernel_times_ms())) / 1e3, d.program_stats(True)[0]))
        print(f"L={L} K={K} leaves={res[0][2]}: groups {res[0][1]:.1f} M  lane {res[1][1]:.1f} M ({res[1][0]})", flush=True)
