"""Lane-per-path list kernel against the path-group kernels over code rates (diagnostic build: PCD_FORCE_LANE / PCD_NO_LANE at
decoder creation), N = 1024, L = 8, CRC-8: where the host's rule (forking leaves) should put the line.  Run on the GPU box."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PYPOLAR_AMD_LIB"] = os.path.join(R, "polar-codes_amd/lib/libpolar_amd_diag.so")
sys.path.insert(0, os.path.join(R, "polar-codes_amd"))
import torch, pypolar_amd
from pypolar_amd.construction import frozen_bits
B = 1 << 18
DT = sys.argv[1] if len(sys.argv) > 1 else "float"  # "char": the int8 decoder on int8 rows
for K, kind in ((32, "BB"), (64, "BB"), (100, "BB"), (176, "5G"), (256, "BB"), (400, "BB"), (512, "BB"), (640, "BB"), (768, "BB"), (900, "BB"), (980, "BB")):
    F = frozen_bits(1024, K, 0.0, kind)
    x = torch.randn(B, 1024, device="cuda") * 2 + 1
    if DT == "char": x = torch.clamp(torch.round(x * 8), -128, 127).to(torch.int8)
    res = []
    for sw in ("PCD_FORCE_LANE", "PCD_NO_LANE"):
        for k in ("PCD_FORCE_LANE", "PCD_NO_LANE"): os.environ.pop(k, None)
        os.environ[sw] = "1"
        d = pypolar_amd.PolarDecoder(1024, 8, F, DT); d.setErrorDetection(8, "crc")
        lanes = d.launch_info(True)["path_groups"] == 64
        d.decode_batch(x); torch.cuda.synchronize(); t = time.time()
        for _ in range(3): d.decode_batch(x)
        torch.cuda.synchronize()
        res.append((lanes, B * 3 / (time.time() - t) / 1e6))
    st = d.program_stats(True) if hasattr(d, "program_stats") else None
    print(f"K={K:4d} {kind}: lane {'yes' if res[0][0] else 'no '} {res[0][1]:7.2f} M cw/s | groups {res[1][1]:7.2f} M cw/s   {st}", flush=True)
