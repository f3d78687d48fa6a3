"""List decoder rate and oracle agreement for the codewords-per-wave choices (diagnostic build: PCD_CWL).
usage: PCD_CWL=<n> python tools/list_cwl_try.py L"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))
import numpy as np, torch, bench, pypolar_amd, oracle_binding as ob
from pypolar_amd.construction import frozen_bits
L = int(sys.argv[1])
dev = torch.device("cuda", 0)
for N in (tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (64, 128, 256, 512, 1024, 2048)):
    K = N // 2
    cfg = dict(N=N, K=K, L=L, dsnr=0.0, crc=8, crc_type="crc", ebn0=2.0)
    fz = frozen_bits(N, K, 0.0, "BB")
    B = 1 << (17 if N <= 2048 else 14)
    llr, info = bench.make_batch(cfg, fz, B, dev, seed=5)
    d = pypolar_amd.PolarDecoder(N, L, fz, "float"); d.setErrorDetection(8, "crc")
    out, ok, met = d.decode_batch(llr, return_metrics=True)
    h = llr[:256].cpu().numpy()
    e_out, e_ok, e_met = ob.decode_batch(N, L, np.asarray(fz, np.uint32), 8, h, True, "float")
    good = np.array_equal(out[:256].cpu().numpy(), e_out) and np.array_equal(met[:256].cpu().numpy(), e_met)
    d.set_timing(True)
    for _ in range(4): d.decode_batch(llr)
    torch.cuda.synchronize()
    ms = float(np.mean(d.kernel_times_ms()[-4:]))
    print(f"L={L} N={N}: {B / ms / 1e3:.1f} M cw/s  oracle {'ok' if good else 'MISMATCH'}", flush=True)
