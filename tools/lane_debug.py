"""Debug aid for the lane-per-path list kernel: cfg 3 / cfg 4 fixtures through the diagnostic library, outputs compared
with the reference's (fixture) results; on a mismatch the per-leaf trace of the first bad row is compared with the oracle's."""
import os, sys
_R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(_R, 'polar-codes_amd'), os.path.join(_R, 'tests')]
os.environ.setdefault('PYPOLAR_AMD_LIB', os.path.join(_R, 'polar-codes_amd', 'lib', 'libpolar_amd_diag.so'))
import numpy as np
import golden_io as gio
import oracle_binding as ob
import pypolar_amd

names = sys.argv[1:] or ["cfg3_n1024_k512_l8_crc24_awgn.bin", "cfg4_n1024_k176_l8_crc11_awgn.bin", "cfg3_n1024_k512_l8_crc24_lowsnr.bin",
                         "cfg3_n1024_k512_l8_crc24_noise.bin", "cfg3_n1024_k512_l8_crc24_quant.bin", "cfg3_n1024_k512_l8_crc24_ties.bin"]
bad_total = 0
for name in names:
    for fx in gio.load(os.path.join(gio.GOLDEN_DIR, name)):
        d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype)
        d.setErrorDetection(fx.crc_size, fx.crc_type)
        d.setSystematic(fx.systematic)
        out, ok, met = d.decode_batch(fx.llr, return_metrics=True)
        bo = np.where((out != fx.out).any(axis=1))[0]
        bk = np.where(ok != fx.ok)[0]
        bm = np.where((met.view(np.uint32) != fx.metrics.view(np.uint32)).any(axis=1))[0]
        print(f"{fx.name}: rows={fx.count} bad_out={len(bo)} bad_ok={len(bk)} bad_metrics={len(bm)}", flush=True)
        bad = sorted(set(bo) | set(bk) | set(bm))
        bad_total += len(bad)
        if bad:
            r = bad[0]
            print("  first bad row", r, "ok", ok[r], fx.ok[r])
            print("  metrics got", met[r])
            print("  metrics ref", fx.metrics[r])
            tr = d.scl_trace(fx.llr[r])
            et = ob.scl_trace(fx.N, fx.L, fx.frozen, fx.llr[r])
            n = et.shape[0]
            for i in range(n):
                if not np.array_equal(tr[i].view(np.uint32), et[i].view(np.uint32)):
                    print("  first differing leaf", i, "of", n)
                    for k in range(max(0, i - 1), min(n, i + 2)):
                        print("   got", k, tr[k])
                        print("   ref", k, et[k])
                    break
            else:
                print("  traces equal over", n, "leaves")
print("TOTAL BAD", bad_total)
sys.exit(1 if bad_total else 0)
