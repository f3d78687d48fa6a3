"""Two-lanes-per-codeword SC kernel against sc_decode_kernel on the same rows (diagnostic build: PCD_NO_PAIR picks the old kernel).
python3 tools/pair_debug.py [log2_rows] -- run on the GPU box."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PYPOLAR_AMD_LIB"] = os.path.join(R, "polar-codes_amd/lib/libpolar_amd_diag.so")
sys.path.insert(0, os.path.join(R, "polar-codes_amd")); sys.path.insert(0, R)
import numpy as np, torch
import pypolar_amd
from pypolar_amd.construction import frozen_bits

def run(dec, llr, reps=1):
    out, ok = dec.decode_batch(llr)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        out, ok = dec.decode_batch(llr)
    torch.cuda.synchronize()
    return out.cpu().numpy(), ok.cpu().numpy(), (time.time() - t) / reps

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cases = [("cfg2", 1024, 512, 0.0, True, 0, "crc"), ("cfg2 crc8", 1024, 512, 0.0, True, 8, "crc"), ("nonsys", 1024, 512, 0.0, False, 8, "crc"),
         ("K=256", 1024, 256, 1.0, True, 8, "crc"), ("K=768", 1024, 768, 0.0, True, 32, "crc"), ("K=100", 1024, 100, 0.0, True, 8, "crc"),
         ("K=900", 1024, 900, 2.0, True, 8, "crc"), ("K=333", 1024, 333, -1.0, False, 32, "crc")]
torch.manual_seed(5)
if len(sys.argv) > 2: cases = [c for c in cases if c[0] in sys.argv[2:]]
for name, N, K, dsnr, systematic, crc, ct in cases:
    F = frozen_bits(N, K, dsnr, "BB")
    B = 1 << lg
    llr = (torch.randn(B, N, device="cuda") * 2.0 + 1.0)
    llr[: B // 4] = torch.round(llr[: B // 4] * 2.0) / 2.0  # ties
    res = {}
    for tag, env in (("pair", None), ("old", "1")):
        if env: os.environ["PCD_NO_PAIR"] = env
        else: os.environ.pop("PCD_NO_PAIR", None)
        d = pypolar_amd.PolarDecoder(N, 1, F, "float")
        d.setSystematic(systematic)
        d.setErrorDetection(crc, ct)
        res[tag] = run(d, llr, 3) + (d.launch_info(False)["codewords_per_workgroup"],)
    a, b = res["pair"], res["old"]
    bad = np.nonzero((a[0] != b[0]).any(axis=1))[0]
    print(f"{name:10s} cw/wg {a[3]:2d}/{b[3]:2d}  rows differ {len(bad)}/{B}  ok differ {(a[1] != b[1]).sum()}  "
          f"pair {B / a[2] / 1e6:8.1f} M cw/s  old {B / b[2] / 1e6:8.1f} M cw/s", flush=True)
    if len(bad):
        r = bad[0]
        x = np.unpackbits(a[0][r]) ^ np.unpackbits(b[0][r])
        print("   first bad row", r, "bit diffs at", np.nonzero(x)[0][:40], "count", x.sum())
