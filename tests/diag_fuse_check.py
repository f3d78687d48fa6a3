"""Child process of tests/test_gpu_parity.py::test_fused_and_unfused_programs_agree: runs with PYPOLAR_AMD_LIB pointing
at the DIAGNOSTIC build of the library and ONE experiment switch in the environment, and checks that the decoders still
reproduce the reference's vectors (and the oracle on random frozen sets).  Exit status 0 = all equal."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "polar-codes_amd"), os.path.join(ROOT, "tests")]
import golden_io as gio  # noqa: E402
import oracle_binding as ob  # noqa: E402
import pypolar_amd  # noqa: E402

METRIC_RTOL = 1e-5
assert pypolar_amd.library_path().endswith("libpolar_amd_diag.so"), pypolar_amd.library_path()


def decoder(fx):
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype)
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    d.setSystematic(fx.systematic)
    return d


names = ["cfg3_n1024_k512_l8_crc24_awgn.bin", "cfg3_n1024_k512_l8_crc24_ties.bin", "cfg4_n1024_k176_l8_crc11_awgn.bin",
         "mix_n256_k128_l4_crc8_awgn.bin", "sweep_n512.bin", "cfg2_n1024_k512_sc_awgn.bin", "cfg2_n1024_k512_sc_ties.bin",
         "sweep_n2048.bin", "cfg1_n64_k40_sc_awgn.bin", "nsys_cfg3_n1024_k512_l8_crc24.bin"]
for name in names:
    for fx in gio.load(os.path.join(gio.GOLDEN_DIR, name)):
        if fx.dectype != "float":
            continue
        res = decoder(fx).decode_batch(fx.llr, return_metrics=fx.L > 1)
        assert np.array_equal(res[0], fx.out), fx.name
        assert np.array_equal(res[1].astype(np.uint8), fx.ok), fx.name
        if fx.L > 1:
            for r in range(fx.count):
                p = int(fx.npaths[r])
                g, e = res[2][r, :p].astype(np.float64), fx.metrics[r, :p].astype(np.float64)
                assert np.all(np.abs(g - e) <= METRIC_RTOL * np.maximum(1.0, np.abs(e))), (fx.name, r)
rng = np.random.default_rng(5)
for N, L in [(256, 2), (256, 4), (512, 8), (128, 8), (256, 1), (2048, 1), (1024, 8)]:
    K = N // 2
    frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    llr = (rng.standard_normal((48, N)) * 2.0 + 1.0).astype(np.float32)
    try:
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, llr, True, "float")
    except ValueError:
        continue  # (a frozen set the reference's tree builder refuses)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    d.setErrorDetection(8, "crc")
    res = d.decode_batch(llr, return_metrics=L > 1)
    assert np.array_equal(res[0], e_out) and np.array_equal(res[1].astype(np.uint8), e_ok), (N, L)
    if L > 1:
        assert np.allclose(res[2], e_met, rtol=METRIC_RTOL, atol=0), (N, L)
print("ok")
