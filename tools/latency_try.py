import os, sys, time
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
import numpy as np, torch, golden_io as gio, pypolar_amd
for name in ("cfg2_n1024_k512_sc_awgn.bin", "cfg3_n1024_k512_l8_crc24_awgn.bin"):
    fx = gio.load_named(name)
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype)
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    res = []
    for B in [int(b) for b in os.environ.get("LAT_B", "1,4,8,16,32,64").split(",")]:
        x = np.ascontiguousarray(np.tile(fx.llr, ((B + fx.count - 1) // fx.count, 1))[:B]); xd = torch.from_numpy(x).cuda()
        for _ in range(5): d.decode_batch(xd)
        torch.cuda.synchronize(); td = []
        for _ in range(200):
            t0 = time.perf_counter(); d.decode_batch(xd); torch.cuda.synchronize(); td.append(time.perf_counter() - t0)
        res.append(f"B={B}: {np.median(td)*1e6:.0f}")
    print(os.environ.get('TAG',''), name[:4], d.kernel_name(fx.L > 1), d.launch_info(fx.L>1)['codewords_per_workgroup'], "us/call:", " ".join(res), flush=True)
