"""GPU suite (-m gpu): the HIP decoders, called through the C ABI (pypolar_amd -> libpolar_amd.so),
against (a) the committed golden vectors of the unmodified reference, (b) the oracle on seeded
inputs, (c) size-independent properties at BASELINE.json's full batch sizes.

Bar: decoded bytes and detector verdicts bit-exact; path metrics within 1e-5 relative (north star)
-- in practice they are bit-identical and the test also reports that."""
import os

import numpy as np
import pytest

import golden_io as gio
import oracle_binding as ob

pytestmark = pytest.mark.gpu

FILES = gio.all_files()
METRIC_RTOL = 1e-5


def _decoder(fx, device=0):
    import pypolar_amd
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype, device=device)
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    d.setSystematic(fx.systematic)
    return d


def _rows(fx):
    """the fixture's rows as the reference received them: float32, or int8 through decode_vector(const char*)"""
    return fx.llr.astype(np.int8) if (fx.i8_input or fx.dectype == "char") else fx.llr


def _check_metrics(got, ref, npaths, name):
    for r in range(ref.shape[0]):
        p = int(npaths[r])
        g, e = got[r, :p].astype(np.float64), ref[r, :p].astype(np.float64)
        assert np.all(np.abs(g - e) <= METRIC_RTOL * np.maximum(1.0, np.abs(e))), (name, r, g, e)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_hip_matches_reference_vectors(built_library, path):
    for fx in gio.load(path):
        d = _decoder(fx)
        want_m = fx.L > 1 and fx.dectype in ("float", "char")
        res = d.decode_batch(_rows(fx), return_metrics=want_m)
        assert np.array_equal(res[0], fx.out), fx.name
        assert np.array_equal(res[1].astype(np.uint8), fx.ok), fx.name
        if want_m:
            _check_metrics(res[2], fx.metrics, fx.npaths, fx.name)


def test_metrics_are_bit_identical_on_cfg3(built_library):
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    _, _, met = _decoder(fx).decode_batch(fx.llr, return_metrics=True)
    assert np.array_equal(met.view(np.uint32), fx.metrics.view(np.uint32))


def test_decode_vector_surface(built_library):
    """pypolar.PolarDecoder.decode_vector drop-in: README example + qa_pypolar_decoder.py:94-132"""
    import pypolar_amd
    fx = gio.load_named("cfg1_n64_k40_sc_awgn.bin")
    d = pypolar_amd.PolarDecoder(64, 1, fx.frozen, "float")
    d.setErrorDetection(0, "crc")
    for r in range(8):
        out = d.decode_vector(fx.llr[r])
        assert out.dtype == np.uint8 and out.shape == (5,)
        assert np.array_equal(out, fx.out[r])
        # float64 / python lists are force-cast to float32 like pybind11's forcecast
        assert np.array_equal(d.decode_vector(fx.llr[r].astype(np.float64).tolist()), fx.out[r])


def test_n8_known_answer(built_library):
    """decodingtest.cpp:1143-1173: N=8, frozen {0,1,2,4}, SCL-float L=4 and SC => high nibble 0xF"""
    import pypolar_amd
    llr = np.array([-5, -6, -4, 1, -4, -5, -7, 2], np.float32)
    for L in (1, 4):
        d = pypolar_amd.PolarDecoder(8, L, [0, 1, 2, 4], "float")
        d.setErrorDetection(0, "crc")
        out, _ = d.decode_batch(llr.reshape(1, 8))
        assert (out[0, 0] & 0xF0) == 0xF0


@pytest.mark.parametrize("name", ["cfg2_n1024_k512_sc_quant.bin", "cfg3_n1024_k512_l8_crc24_quant.bin",
                                  "cfg4_n1024_k176_l8_crc11_quant.bin"])
def test_int8_input_path(built_library, name):
    """decode_vector(np.int8): int8 LLRs widened to float (bitcontainer.cpp:202-207). The quant fixtures
    hold integer-valued float LLRs, so the int8 overload must give the reference's float result."""
    fx = gio.load_named(name)
    d = _decoder(fx)
    q = fx.llr.astype(np.int8)
    assert np.array_equal(q.astype(np.float32), fx.llr)
    out, ok = d.decode_batch(q)
    assert np.array_equal(out, fx.out) and np.array_equal(ok.astype(np.uint8), fx.ok)
    assert np.array_equal(d.decode_vector(q[0]), fx.out[0, : fx.K // 8])


@pytest.mark.parametrize("N,L,det", [(64, 1, 0), (256, 1, 8), (256, 4, 8), (1024, 8, 24), (512, 32, 32), (128, 16, 8),
                                     (2048, 2, 11)])
def test_random_inputs_vs_oracle(built_library, N, L, det):
    """seeded random inputs (pure noise, low SNR, ties) the fixtures do not contain, systematic and
    non-systematic output, checked row by row against the oracle."""
    import pypolar_amd
    rng = np.random.default_rng(N * 131 + L * 7 + det)
    src = {64: "cfg1_n64_k40_sc_awgn.bin", 256: "sweep_n256.bin", 1024: "cfg3_n1024_k512_l8_crc24_awgn.bin",
           512: "sweep_n512.bin", 128: "sweep_n128.bin", 2048: "sweep_n2048.bin"}[N]
    cands = [f for f in gio.load(os.path.join(gio.GOLDEN_DIR, src))]
    frozen = cands[len(cands) // 2].frozen
    K = N - frozen.size
    if det and K // 8 < 6:
        det = 0
    B = 48 if N <= 512 else 24
    rows = []
    for i in range(B):
        mode = i % 3
        if mode == 0:
            rows.append(rng.normal(0, 1, N))
        elif mode == 1:
            info = rng.integers(0, 256, (K + 7) // 8, dtype=np.uint8)
            if K % 8:
                info[-1] &= (0xFF << (8 - K % 8)) & 0xFF
            _, cw = ob.encode(N, frozen, det, info)
            rows.append(1.0 - 2.0 * np.unpackbits(cw)[:N] + rng.normal(0, 0.9, N))
        else:
            rows.append(rng.choice([-1.0, 1.0], N) * rng.choice([1.0, 2.0], N))
    llr = np.asarray(rows, np.float32)
    for systematic in (True, False):
        d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
        d.setErrorDetection(det, "crcnr" if det in (11, 24) else "crc")
        d.setSystematic(systematic)
        res = d.decode_batch(llr, return_metrics=L > 1)
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr, systematic, "float")
        assert np.array_equal(res[0], e_out), (N, L, det, systematic)
        assert np.array_equal(res[1].astype(np.uint8), e_ok)
        if L > 1:
            assert np.allclose(res[2], e_met, rtol=METRIC_RTOL, atol=0)


def test_mixed_decoder_vs_oracle(built_library):
    import pypolar_amd
    fx = gio.load_named("mix_n1024_k512_l8_crc24_awgn.bin")
    rng = np.random.default_rng(5)
    llr = np.concatenate([fx.llr, rng.normal(0, 1, (16, fx.N)).astype(np.float32)])
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, "mixed")
    d.setErrorDetection(24, "crcnr")
    out, ok = d.decode_batch(llr)
    e_out, e_ok, _ = ob.decode_batch(fx.N, fx.L, fx.frozen, 24, llr, True, "mixed")
    assert np.array_equal(out, e_out) and np.array_equal(ok.astype(np.uint8), e_ok)
    # int8 rows (decode_vector(const char*)): the SC pass reads them as they are, the list pass widens the rows it was handed
    rows8 = np.clip(np.rint(llr * 4.0), -128, 127).astype(np.int8)
    for B in (rows8.shape[0], 37, 1):
        out8, ok8 = d.decode_batch(rows8[:B])
        e_out8, e_ok8, _ = ob.decode_batch(fx.N, fx.L, fx.frozen, 24, rows8[:B].astype(np.float32), True, "mixed")
        assert np.array_equal(out8, e_out8) and np.array_equal(ok8.astype(np.uint8), e_ok8)
        assert 0 < int(e_ok8.sum()) or B == 1


def test_device_pointer_path_and_ragged_batches(built_library):
    """torch CUDA tensors go through PCD_MEM_DEVICE (no staging); batch sizes that do not divide the
    grid, a single row and an empty batch."""
    import torch
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    d = _decoder(fx)
    dev = torch.device("cuda:0")
    for B in (0, 1, 3, 65, fx.count):
        x = torch.from_numpy(fx.llr[:B].copy()).to(dev)
        out, ok = d.decode_batch(x)
        torch.cuda.synchronize()
        assert out.shape == (B, 64)
        assert np.array_equal(out.cpu().numpy(), fx.out[:B])
        assert np.array_equal(ok.cpu().numpy(), fx.ok[:B])


def _systematic_codewords(N, frozen, det, B, rng):
    """B valid systematic codewords (oracle encoder) -> (info_with_crc[B, nb], bits[B, N])"""
    K = N - frozen.size
    infos, cws = [], []
    for _ in range(B):
        info = rng.integers(0, 256, K // 8, dtype=np.uint8)
        i2, cw = ob.encode(N, frozen, det, info)
        infos.append(i2)
        cws.append(np.unpackbits(cw))
    return np.asarray(infos), np.asarray(cws)


FULL_BATCH = [
    # fixture (code), decoder type, list size, detector bits, Eb/N0 dB, log2 batch
    ("cfg2_n1024_k512_sc_awgn.bin", "float", 1, 0, 4.0, 20),
    ("cfg3_n1024_k512_l8_crc24_awgn.bin", "float", 8, 24, 3.0, 20),
    ("cfg4_n1024_k176_l8_crc11_awgn.bin", "float", 8, 11, 1.0, 19),
    ("cfg5_n4096_k2048_l32_crc32_awgn.bin", "float", 32, 32, 2.5, 15),
    ("cfg3_n1024_k512_l8_crc24_awgn.bin", "char", 8, 24, 3.0, 20),
    ("cfg2_n1024_k512_sc_awgn.bin", "char", 1, 0, 4.0, 20),
    ("cfg3_n1024_k512_l8_crc24_awgn.bin", "mixed", 8, 24, 1.75, 20),
]


@pytest.mark.parametrize("name,dectype,L,det,ebn0,lb", FULL_BATCH, ids=[f"{n.split('_')[0]}_{t}_2^{lb}" for n, t, _, _, _, lb in FULL_BATCH])
def test_full_batch_properties(built_library, name, dectype, L, det, ebn0, lb):
    """BASELINE batch sizes (device resident): encode -> AWGN -> decode round trip on a tiled set of distinct codewords;
    every replica of a row must decode identically (determinism across waves, grid-stride iterations and arena reuse),
    rows that are decoded must return their payload with ok = 1, and a sample of rows must equal the oracle (path metrics
    included for the list decoders).  The int8 decoders get the LLRs scaled and rounded to int8; "mixed" runs near the SC
    failure threshold, so that the list pass sees a long compacted row list."""
    import pypolar_amd
    import torch
    fx = gio.load_named(name)
    N, frozen = fx.N, fx.frozen
    K = N - frozen.size
    rng = np.random.default_rng(99)
    U = 512
    dname = "crcnr" if det in (11, 24) else "crc"
    infos, bits = _systematic_codewords(N, frozen, det, U, rng)
    sigma = 1.0 / np.sqrt(2.0 * 10 ** (ebn0 / 10.0) * K / N)
    llr_u = (1.0 - 2.0 * bits + sigma * rng.normal(0, 1, bits.shape)).astype(np.float32)
    if dectype == "char":
        llr_u = np.clip(np.rint(llr_u * 16.0), -128, 127).astype(np.int8)
    B = 1 << lb
    dev = torch.device("cuda:0")
    x = torch.from_numpy(llr_u).to(dev).repeat(B // U, 1)
    d = pypolar_amd.PolarDecoder(N, L, frozen, dectype)
    d.setErrorDetection(det, dname)
    want_m = L > 1 and dectype != "mixed"
    res = d.decode_batch(x, return_metrics=want_m)
    torch.cuda.synchronize()
    out = res[0].view(B // U, U, -1)
    assert bool((out == out[0:1]).all()), "replicas of the same row decoded differently"
    ok = res[1].view(B // U, U)
    assert bool((ok == ok[0:1]).all())
    if want_m:
        met = res[2].view(B // U, U, -1)
        assert bool((met.view(torch.int32) == met[0:1].view(torch.int32)).all()), "replicas of the same row carry different metrics"
    o0, k0 = out[0].cpu().numpy(), ok[0].cpu().numpy()
    S = 64
    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr_u[:S].astype(np.float32), True, dectype)
    assert np.array_equal(o0[:S], e_out) and np.array_equal(k0[:S], e_ok)
    if want_m:
        m0 = res[2].view(B // U, U, -1)[0, :S].cpu().numpy()
        assert np.allclose(m0, e_met, rtol=METRIC_RTOL, atol=0)
    good = (o0 == infos).all(axis=1)
    assert good.mean() > (0.5 if dectype == "mixed" or name.startswith("cfg4") else 0.97), good.mean()
    if det:
        assert np.array_equal(k0.astype(bool) & good, good)


FULL_BATCH_R4 = [
    # N, K, decoder type, list size, detector bits, Eb/N0 dB, log2 batch -- the kernels of round 4 at batch sizes that take every wave through many groups
    (128, 64, "float", 1, 0, 5.0, 20), (256, 128, "char", 1, 0, 5.0, 20), (512, 256, "float", 1, 8, 4.0, 20),      # sc_pair_n7 / n8 / n9_kernel
    (64, 32, "float", 1, 0, 5.0, 20), (64, 32, "char", 1, 8, 5.0, 20),                                                # sc_pair_n6_kernel
    (64, 32, "char", 32, 8, 4.0, 16), (128, 64, "float", 8, 8, 4.0, 19), (256, 128, "float", 2, 8, 4.0, 20),          # lane kernels at N <= 256
    (1024, 512, "float", 4, 8, 3.0, 18), (2048, 1024, "float", 2, 8, 3.0, 17), (512, 256, "char", 4, 8, 3.5, 18),     # 2 / 4 lanes per codeword
]


@pytest.mark.parametrize("N,K,dectype,L,det,ebn0,lb", FULL_BATCH_R4, ids=[f"N{n}_L{l}_{t}_2^{lb}" for n, _, t, l, _, _, lb in FULL_BATCH_R4])
def test_full_batch_properties_of_the_round4_kernels(built_library, N, K, dectype, L, det, ebn0, lb):
    """the same properties as above for the kernels that round 4 added (pair kernels at N <= 512, lane kernels at N <= 256 and with 2 / 4
    lanes per codeword): every replica of a row decodes identically, decoded rows return their payload, a sample equals the oracle"""
    import pypolar_amd
    import torch
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
    rng = np.random.default_rng(1234 + N + L)
    U = 512
    infos, bits = _systematic_codewords(N, frozen, det, U, rng)
    sigma = 1.0 / np.sqrt(2.0 * 10 ** (ebn0 / 10.0) * K / N)
    llr_u = (1.0 - 2.0 * bits + sigma * rng.normal(0, 1, bits.shape)).astype(np.float32)
    if dectype == "char":
        llr_u = np.clip(np.rint(llr_u * 16.0), -128, 127).astype(np.int8)
    B = 1 << lb
    x = torch.from_numpy(llr_u).to(torch.device("cuda:0")).repeat(B // U, 1)
    d = pypolar_amd.PolarDecoder(N, L, frozen, dectype)
    d.setErrorDetection(det, "crc")
    assert d.kernel_name(L > 1).startswith("sc_pair_n" if L == 1 else "scl_lane_n"), d.kernel_name(L > 1)
    res = d.decode_batch(x, return_metrics=L > 1)
    torch.cuda.synchronize()
    out = res[0].view(B // U, U, -1)
    assert bool((out == out[0:1]).all()), "replicas of the same row decoded differently"
    ok = res[1].view(B // U, U)
    assert bool((ok == ok[0:1]).all())
    if L > 1:
        met = res[2].view(B // U, U, -1)
        assert bool((met.view(torch.int32) == met[0:1].view(torch.int32)).all()), "replicas of the same row carry different metrics"
    o0, k0 = out[0].cpu().numpy(), ok[0].cpu().numpy()
    S = 64
    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr_u[:S].astype(np.float32), True, dectype)
    assert np.array_equal(o0[:S], e_out) and np.array_equal(k0[:S], e_ok)
    if L > 1:
        assert np.array_equal(res[2].view(B // U, U, -1)[0, :S].cpu().numpy().view(np.uint32), e_met.view(np.uint32))
    good = (o0 == infos).all(axis=1)
    assert good.mean() > 0.9, good.mean()
    if det:
        assert np.array_equal(k0.astype(bool) & good, good)


def test_repeated_decodes_are_bit_identical(built_library):
    """the same 2^16 noisy rows decoded five times: bytes, verdicts and path metrics must not change from call to call (a
    counted wait that lets a load race its consumer shows up here as a handful of differing rows -- the split first pass
    over the channel row of round 3 had one)"""
    import pypolar_amd
    import torch
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_lowsnr.bin")
    rng = np.random.default_rng(3)
    base = np.tile(fx.llr, (2048 // fx.count + 1, 1))[:2048]
    x = torch.from_numpy((base + 0.25 * rng.standard_normal(base.shape)).astype(np.float32)).cuda().repeat(32, 1)
    d = _decoder(fx)
    ref = None
    for _ in range(5):
        out, ok, met = d.decode_batch(x, return_metrics=True)
        torch.cuda.synchronize()
        cur = (out.clone(), ok.clone(), met.view(torch.int32).clone())
        if ref is None:
            ref = cur
        else:
            assert all(bool((a == b).all()) for a, b in zip(ref, cur))
    o = ref[0].view(32, 2048, -1)
    assert bool((o == o[0:1]).all())


def test_two_handles_on_two_streams(built_library):
    """two decoders (list and SC) working concurrently on two streams of one device: each owns its scratch (arena, row
    lists), so the results must be those of the sequential calls"""
    import pypolar_amd
    import torch
    dev = torch.device("cuda:0")
    fa = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    fb = gio.load_named("cfg2_n1024_k512_sc_awgn.bin")
    da, db = _decoder(fa), _decoder(fb)
    da2 = _decoder(fa)  # a second list decoder of the same code: its own arena
    reps = 256
    xa = torch.from_numpy(fa.llr).to(dev).repeat(reps, 1)
    xb = torch.from_numpy(fb.llr).to(dev).repeat(reps, 1)
    sa, sb, sc = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        ra = da.decode_batch(xa, return_metrics=True, stream=sa.cuda_stream)
        rb = db.decode_batch(xb, stream=sb.cuda_stream)
        rc = da2.decode_batch(xa, return_metrics=True, stream=sc.cuda_stream)
        outs.append((ra, rb, rc))
    torch.cuda.synchronize()
    for ra, rb, rc in outs:
        for r in (ra, rc):
            assert np.array_equal(r[0].view(reps, fa.count, -1)[7].cpu().numpy(), fa.out)
            assert bool((r[0].view(reps, fa.count, -1) == r[0].view(reps, fa.count, -1)[0:1]).all())
            assert np.array_equal(r[2].view(reps, fa.count, -1)[reps - 1].cpu().numpy().view(np.uint32), fa.metrics.view(np.uint32))
        assert np.array_equal(rb[0].view(reps, fb.count, -1)[3].cpu().numpy(), fb.out)
        assert bool((rb[0].view(reps, fb.count, -1) == rb[0].view(reps, fb.count, -1)[0:1]).all())


LANE_CODES = [(1024, 512, 8, "BB"), (1024, 176, 8, "5G"), (1024, 800, 8, "BB"), (1024, 940, 6, "BE"), (1024, 64, 8, "BB"),
              (1024, 512, 5, "rand"), (1024, 300, 7, "rand"), (1024, 1000, 8, "BB"), (1024, 24, 8, "BB")]


@pytest.mark.parametrize("N,K,L,kind", LANE_CODES, ids=[f"N{n}_K{k}_L{l}_{c}" for n, k, l, c in LANE_CODES])
def test_lane_per_path_kernel_codes(built_library, N, K, L, kind):
    """The lane-per-path list kernel (N = 1024, 4 < L <= 8) over rates, list sizes and frozen sets -- constructions, the 5G
    table and arbitrary sets (whatever tree the lowering cannot express falls back to the path-group kernels: the
    results must be the reference's either way) -- on AWGN, pure-noise and tie-heavy rows, systematic and not."""
    import pypolar_amd
    rng = np.random.default_rng(N + K + L)
    if kind == "rand":
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    else:
        frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0 if kind != "5G" else 0.0, kind), dtype=np.uint32)
    det = 8 if K >= 16 else 0
    rows = 40
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))).astype(np.float32) * np.where(rng.random((rows, N)) < 0.5, 1, -1).astype(np.float32)
    noise = rng.standard_normal((16, N)).astype(np.float32)
    ties = rng.integers(-3, 4, (24, N)).astype(np.float32)
    llr = np.concatenate([awgn, noise, ties])
    for systematic in (True, False):
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr, systematic, "float")
        d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
        d.setErrorDetection(det, "crc")
        d.setSystematic(systematic)
        out, ok, met = d.decode_batch(llr, return_metrics=True)
        assert np.array_equal(out, e_out), (N, K, L, kind, systematic)
        assert np.array_equal(ok.astype(np.uint8), e_ok)
        assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), "path metrics differ from the oracle's"


LANE_GEO_CODES = [(512, 256, 8, "BB"), (512, 128, 6, "BB"), (512, 256, 16, "BB"), (512, 200, 32, "BE"), (512, 256, 5, "rand"),
                  (1024, 512, 16, "BB"), (1024, 300, 12, "BE"), (1024, 512, 32, "BB"), (1024, 800, 24, "BB"), (1024, 512, 9, "rand"),
                  (2048, 1024, 8, "BB"), (2048, 512, 7, "BE"), (2048, 1500, 8, "BB"), (2048, 1024, 16, "BB"), (2048, 700, 24, "BB"), (2048, 1024, 8, "rand"),
                  (4096, 2048, 8, "BB"), (4096, 2048, 16, "BB"), (4096, 2048, 32, "BB"), (4096, 2048, 20, "BE"),
                  # low rates: an all-frozen leaf of 512 / 1024 LLRs, recomputed from the channel row (the *_hi kernels)
                  (4096, 1400, 32, "BB"), (2048, 256, 8, "BB"), (4096, 512, 16, "BB"), (4096, 256, 8, "BE"), (2048, 128, 32, "BB"),
                  (2048, 72, 17, "BB"), (2048, 72, 8, "BB"),      # (an all-frozen leaf of 1024 LLRs at N = 2048: found by the round-4 fuzz)
                  # high rates: RateOne / SPC leaves of 512 / 1024 LLRs (scan sink of the pass, sign words rewritten in place)
                  (2048, 1664, 8, "BB"), (2048, 1920, 16, "BB"), (2048, 1800, 32, "BE"), (4096, 2304, 8, "BB"), (4096, 2816, 16, "BB"),
                  (4096, 3328, 32, "BB"), (4096, 3840, 8, "BE"), (4096, 3000, 17, "BB"),
                  # 2 / 4 lanes per codeword (list sizes 2, 3 .. 4: 32 / 16 codewords per wave)
                  (512, 256, 2, "BB"), (512, 300, 4, "BE"), (1024, 512, 2, "BB"), (1024, 512, 4, "BB"), (1024, 700, 3, "BE"), (1024, 512, 4, "rand"),
                  (2048, 1024, 2, "BB"), (2048, 1024, 4, "BB"), (2048, 300, 3, "BB"), (2048, 1800, 2, "BE"), (4096, 2048, 4, "BB"), (4096, 3000, 3, "BB"),
                  (4096, 1100, 4, "BE"),
                  # N = 256: stage 8 is the channel row (no top-of-tree pass; the root's children are register blocks)
                  (256, 128, 8, "BB"), (256, 128, 2, "BB"), (256, 100, 3, "BE"), (256, 64, 16, "BB"), (256, 200, 32, "BB"), (256, 128, 5, "rand"), (256, 230, 8, "BB"),
                  # N = 128: stage 7 (the LDS column of the first path) is the channel row
                  (128, 64, 8, "BB"), (128, 64, 2, "BB"), (128, 50, 4, "BE"), (128, 32, 16, "BB"), (128, 100, 32, "BB"), (128, 64, 7, "rand"), (128, 8, 8, "BB"),
                  # N = 64: stage 6 (a register block, copied into every lane)
                  (64, 32, 8, "BB"), (64, 32, 2, "BB"), (64, 24, 4, "BE"), (64, 16, 16, "BB"), (64, 50, 32, "BB"), (64, 32, 6, "rand")]


@pytest.mark.parametrize("N,K,L,kind", LANE_GEO_CODES, ids=[f"N{n}_K{k}_L{l}_{c}" for n, k, l, c in LANE_GEO_CODES])
def test_lane_kernel_geometries(built_library, N, K, L, kind):
    """The lane-per-path list kernel under its other geometries (round 4, csrc/scl_lane_geo.h): N = 512 / 2048 / 4096 -- every stage
    between the channel row and stage 8 recomputed by the generic top-of-tree pass, the path-independent ones shared through the
    virtual rows -- and 16 / 32 lanes per codeword (distributed merge selection, near ties resolved on the exact metrics), on AWGN,
    pure-noise and tie-heavy rows, systematic and not, ragged batch sizes; bytes, verdicts and all L path metrics bit for bit.
    Reference: SclAvxFloat scl_avx_float.cpp:229-263, 316-651, 711-750."""
    import pypolar_amd
    rng = np.random.default_rng(7 * N + K + L)
    if kind == "rand":
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
    else:
        frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, kind), dtype=np.uint32)
    rows = (21 if N <= 1024 else (13 if N == 2048 else 7)) + (32 if L <= 4 else 0)     # (not a multiple of the codewords per wave, more than one wave's worth)
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))).astype(np.float32) * np.where(rng.random((rows, N)) < 0.5, 1, -1).astype(np.float32)
    noise = rng.standard_normal((rows // 3, N)).astype(np.float32)
    ties = rng.integers(-3, 4, (rows // 2, N)).astype(np.float32)
    eighths = (np.round(awgn[: rows // 2] * 8) / 8).astype(np.float32)    # quantised: equal metrics at most forks
    llr = np.concatenate([awgn, noise, ties, eighths])
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    if kind in ("BB", "BE"):
        assert d.kernel_name(True).startswith("scl_lane_"), d.kernel_name(True)   # (every construction of this list is expressible: the lane kernels run)
    for systematic in (True, False):
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, llr, systematic, "float")
        d.setErrorDetection(8, "crc")
        d.setSystematic(systematic)
        out, ok, met = d.decode_batch(llr, return_metrics=True)
        assert np.array_equal(out, e_out), (N, K, L, kind, systematic)
        assert np.array_equal(ok.astype(np.uint8), e_ok)
        assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), "path metrics differ from the oracle's"


LANE_GEO_CHAR_CODES = [(512, 200, 32), (1024, 512, 16), (1024, 512, 32), (2048, 1024, 8), (2048, 1024, 16), (2048, 700, 32), (2048, 256, 8), (4096, 1024, 8),
                       (2048, 1800, 8), (4096, 3000, 8), (4096, 2560, 32), (2048, 256, 16),   # (leaves of 512 / 1024 LLRs, the repetition leaf included)
                       (512, 200, 2), (1024, 512, 4), (1024, 512, 2), (2048, 1024, 3), (2048, 600, 2), (4096, 2048, 4), (4096, 3000, 4),   # (2 / 4 lanes per codeword)
                       (256, 128, 8), (256, 128, 2), (256, 200, 16), (256, 64, 32),   # (N = 256)
                       (128, 64, 8), (128, 64, 2), (128, 100, 16), (128, 32, 32),   # (N = 128)
                       (64, 32, 8), (64, 32, 2), (64, 48, 16), (64, 16, 32)]   # (N = 64)


@pytest.mark.parametrize("N,K,L", LANE_GEO_CHAR_CODES, ids=[f"N{n}_K{k}_L{l}" for n, k, l in LANE_GEO_CHAR_CODES])
def test_lane_kernel_geometries_int8_decoder(built_library, N, K, L):
    """the same geometries for the int8 decoder SclFipChar (scl_fip_char.cpp:388-775; scl_lane_n*_l*_char[_hi].hip): int8 rows read as
    bytes and widened in the top-of-tree pass, saturating f / g at every recomputed level, integer metrics (equal at most forks)"""
    import pypolar_amd
    rng = np.random.default_rng(11 * N + K + L)
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
    rows = 13 if N <= 2048 else 6
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))) * np.where(rng.random((rows, N)) < 0.5, 1, -1)
    llr = np.concatenate([np.clip(np.round(awgn * 8), -128, 127), rng.integers(-128, 128, (rows // 2, N)), rng.integers(-3, 4, (rows // 2, N))]).astype(np.int8)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "char")
    assert d.kernel_name(True).startswith("scl_lane_") and "_char" in d.kernel_name(True), d.kernel_name(True)
    for systematic in (True, False):
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, llr.astype(np.float32), systematic, "char")
        d.setErrorDetection(8, "crc")
        d.setSystematic(systematic)
        out, ok, met = d.decode_batch(llr, return_metrics=True)
        assert np.array_equal(out, e_out), (N, K, L, systematic)
        assert np.array_equal(ok.astype(np.uint8), e_ok)
        assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), "path metrics differ from the oracle's"


@pytest.mark.parametrize("dectype,K,kind", [("float", 512, "BB"), ("float", 176, "5G"), ("float", 900, "BB"), ("char", 512, "BB"), ("char", 256, "BB"),
                                            ("char", 768, "BB"), ("char", 900, "BB")],
                         ids=["float_K512", "float_5G_K176", "float_K900_big", "char_K512", "char_K256", "char_K768_big", "char_K900_big"])
def test_lane_kernel_quantised_rows(built_library, dectype, K, kind):
    """Quantised LLRs make equal path metrics the normal case at every fork of every codeword of a wave: the lane kernel's
    three tiers of tie handling (merge order kept / emulation over the candidates that can win / over all of them) must
    reproduce simplePartialSortDescending's swap order (arrayfuncs.h:161-183) -- AWGN rows at 2 dB rounded to 1/8, 1/2 and
    to integers, float decoder (also fed int8 rows: widened onto the lane kernel) and int8 decoder; bytes, verdicts and
    path metrics against the oracle."""
    import pypolar_amd
    N, L = 1024, 8
    rng = np.random.default_rng(K + len(dectype))
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, kind), dtype=np.uint32)
    sigma = 0.9
    y = (1.0 + sigma * rng.standard_normal((96, N))).astype(np.float32) * (2.0 / sigma ** 2)
    sets = [np.round(y * 8) / 8, np.round(y * 2) / 2, np.round(y)]
    if dectype == "char":
        sets = [np.clip(np.rint(y * 8), -128, 127), np.clip(np.rint(y * 2), -128, 127)]
    for llr in sets:
        llr = np.ascontiguousarray(llr, dtype=np.float32)
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, llr, True, dectype)
        d = pypolar_amd.PolarDecoder(N, L, frozen, dectype)
        d.setErrorDetection(8, "crc")
        assert d.launch_info(True)["path_groups"] == 64, "expected the lane-per-path kernel for this code"
        out, ok, met = d.decode_batch(llr, return_metrics=True)
        assert np.array_equal(out, e_out)
        assert np.array_equal(ok.astype(np.uint8), e_ok)
        assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), "path metrics differ from the oracle's"
        if np.abs(llr).max() <= 127 and np.array_equal(llr, np.rint(llr)):  # the same rows as int8 input
            rows8 = llr.astype(np.int8)  # (a rounded -0.0 becomes +0: the oracle gets what the widening delivers)
            e_out8, e_ok8, _ = ob.decode_batch(N, L, frozen, 8, rows8.astype(np.float32), True, dectype)
            out8, ok8 = d.decode_batch(rows8)
            assert np.array_equal(out8, e_out8) and np.array_equal(ok8.astype(np.uint8), e_ok8)


PAIR_CODES = [(512, 0.0, "BB"), (400, 0.0, "BB"), (640, 1.0, "BB"), (176, 0.0, "5G"), (512, 2.0, "BB"), (300, -1.0, "BB"), (768, 0.0, "BB"),
              (256, 1.0, "BB"), (100, 0.0, "BB"), (900, 2.0, "BB"), (512, 0.0, "BE"), (448, 0.5, "BB"), (576, 3.0, "BB")]


@pytest.mark.parametrize("K,dsnr,kind", PAIR_CODES, ids=[f"K{k}_{c}_{d}" for k, d, c in PAIR_CODES])
def test_pair_kernel_codes(built_library, K, dsnr, kind):
    """The two-lanes-per-codeword SC kernel (N = 1024, float) over rates and constructions (the extreme rates run on its
    instantiations with the leaves of 256 / 512 positions) -- whatever tree its lowering cannot express falls back to
    sc_decode_kernel, the results must be the reference's either way -- on AWGN, pure-noise and
    tie-heavy rows, ragged batch sizes (the last wave of 32 codewords partly idle), systematic and not, with and without a
    detector."""
    import pypolar_amd
    N = 1024
    rng = np.random.default_rng(K + int(10 * dsnr) + 7)
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, dsnr, kind), dtype=np.uint32)
    rows = 70
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))).astype(np.float32) * np.where(rng.random((rows, N)) < 0.5, 1, -1).astype(np.float32)
    noise = rng.standard_normal((13, N)).astype(np.float32)
    ties = rng.integers(-3, 4, (22, N)).astype(np.float32)
    half = (rng.integers(-6, 7, (16, N)) / 2.0).astype(np.float32)
    half[half == 0] = -0.0  # (negative zeros among the inputs: the leaf sums start from +0)
    llr = np.concatenate([awgn, noise, ties, half])
    for systematic, det in ((True, 8), (False, 32), (True, 0)):
        e_out, e_ok, _ = ob.decode_batch(N, 1, frozen, det, llr, systematic, "float")
        d = pypolar_amd.PolarDecoder(N, 1, frozen, "float")
        d.setErrorDetection(det, "crc")
        d.setSystematic(systematic)
        if (K, kind) == (512, "BB"):
            assert d.launch_info(False)["codewords_per_workgroup"] == 32, "the headline SC code must run on sc_pair_kernel"
        for cnt in (len(llr), 1, 33):
            out, ok = d.decode_batch(llr[:cnt])
            assert np.array_equal(out, e_out[:cnt]), (K, dsnr, kind, systematic, det, cnt)
            assert np.array_equal(ok.astype(np.uint8), e_ok[:cnt])


PAIR_CODES_512 = [(256, 0.0, "BB"), (200, 0.0, "BB"), (320, 1.0, "BB"), (88, 0.0, "BE"), (256, 2.0, "BB"), (150, -1.0, "BB"), (384, 0.0, "BB"),
                  (128, 1.0, "BB"), (50, 0.0, "BB"), (450, 2.0, "BB"), (256, 0.0, "BE"), (224, 0.5, "BB")]


PAIR_SMALL = [(512, k, d, c) for k, d, c in PAIR_CODES_512] + [(256, 128, 0.0, "BB"), (256, 100, 1.0, "BB"), (256, 200, 0.0, "BE"), (256, 60, -1.0, "BB"),
                                                               (256, 230, 2.0, "BB"), (128, 64, 0.0, "BB"), (128, 40, 1.0, "BE"), (128, 100, 0.0, "BB"), (128, 20, 0.0, "BB"),
                                                               (64, 32, 0.0, "BB"), (64, 20, 1.0, "BE"), (64, 50, 0.0, "BB"), (64, 40, 2.0, "BB")]


@pytest.mark.parametrize("N,K,dsnr,kind", PAIR_SMALL, ids=[f"N{n}_K{k}_{c}_{d}" for n, k, d, c in PAIR_SMALL])
def test_pair_kernel_codes_small_blocks(built_library, N, K, dsnr, kind):
    """The same kernel at N = 512 / 256 / 128 / 64 (sc_pair_n9 / n8 / n7 / n6_kernel, round 4: the channel row is local stage 8 / 7 / 6 / 5 -- fetched
    once per codeword into the AGPRs / the LDS column / the registers, int8 rows widened on the way --; whatever its lowering cannot
    express runs on sc_decode_kernel): rates and constructions, AWGN / noise / tie rows with negative zeros, ragged batches,
    systematic and not, with and without a detector."""
    import pypolar_amd
    rng = np.random.default_rng(K + int(10 * dsnr) + 11)
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, dsnr, kind), dtype=np.uint32)
    rows = 70
    awgn = (1.0 + 0.8 * rng.standard_normal((rows, N))).astype(np.float32) * np.where(rng.random((rows, N)) < 0.5, 1, -1).astype(np.float32)
    noise = rng.standard_normal((13, N)).astype(np.float32)
    ties = rng.integers(-3, 4, (22, N)).astype(np.float32)
    half = (rng.integers(-6, 7, (16, N)) / 2.0).astype(np.float32)
    half[half == 0] = -0.0
    llr = np.concatenate([awgn, noise, ties, half])
    for systematic, det in ((True, 8), (False, 32 if K > 32 else 8), (True, 0)):
        e_out, e_ok, _ = ob.decode_batch(N, 1, frozen, det, llr, systematic, "float")
        d = pypolar_amd.PolarDecoder(N, 1, frozen, "float")
        d.setErrorDetection(det, "crc")
        d.setSystematic(systematic)
        if (kind, dsnr) == ("BB", 0.0) and 2 * K == N:
            assert d.kernel_name(False) == {512: "sc_pair_n9_kernel", 256: "sc_pair_n8_kernel", 128: "sc_pair_n7_kernel", 64: "sc_pair_n6_kernel"}[N]
        for cnt in (len(llr), 1, 33):
            out, ok = d.decode_batch(llr[:cnt])
            assert np.array_equal(out, e_out[:cnt]), (K, dsnr, kind, systematic, det, cnt)
            assert np.array_equal(ok.astype(np.uint8), e_ok[:cnt])
        if det == 8:  # int8 rows into the same decoder (the <false, true> instantiations), and the int8 decoder on them
            rows8 = ties.astype(np.int8)
            e8, eok8, _ = ob.decode_batch(N, 1, frozen, det, rows8.astype(np.float32), systematic, "float")
            o8, ok8 = d.decode_batch(rows8)
            assert np.array_equal(o8, e8) and np.array_equal(ok8.astype(np.uint8), eok8)
            wide = np.clip(np.rint(8.0 * awgn), -128, 127).astype(np.int8)
            rowsc = np.concatenate([rows8, wide])
            ec, eokc, _ = ob.decode_batch(N, 1, frozen, det, rowsc.astype(np.float32), systematic, "char")
            dc = pypolar_amd.PolarDecoder(N, 1, frozen, "char")
            dc.setErrorDetection(det, "crc")
            dc.setSystematic(systematic)
            oc, okc = dc.decode_batch(rowsc)
            assert np.array_equal(oc, ec) and np.array_equal(okc.astype(np.uint8), eokc)


def test_pair_kernel_int8_rows_into_the_float_decoder(built_library):
    """decode_vector(const char*) of the float decoder (the values converted, float arithmetic; fastssc_avx_float.cpp:940-955 via
    FloatContainer::insertCharLlr): int8 rows must decode like the same values as fp32 rows, here on the two-lanes-per-codeword
    kernel's third instantiation (int8 rows fetched as whole lines, float f / g), and like the oracle"""
    import pypolar_amd
    N, K = 1024, 512
    rng = np.random.default_rng(99)
    frozen = np.asarray(pypolar_amd.frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
    rows = 333
    q = np.clip(np.round(8.0 * (1.0 + 0.9 * rng.standard_normal((rows, N)))), -128, 127).astype(np.int8)
    q[:40] = rng.integers(-3, 4, (40, N)).astype(np.int8)
    for systematic, det in ((True, 8), (False, 0)):
        d = pypolar_amd.PolarDecoder(N, 1, frozen, "float")
        d.setErrorDetection(det, "crc")
        d.setSystematic(systematic)
        o8, k8 = d.decode_batch(q)
        of, kf = d.decode_batch(q.astype(np.float32))
        e_out, e_ok, _ = ob.decode_batch(N, 1, frozen, det, q.astype(np.float32), systematic, "float")
        assert np.array_equal(o8, of) and np.array_equal(k8, kf)
        assert np.array_equal(o8, e_out) and np.array_equal(k8.astype(np.uint8), e_ok)


def test_cpp_shim_decodes_on_gpu(built_library):
    """C++ host side (polar_decoder.hpp over the C ABI): N=8 known answer of decodingtest.cpp:1143-1173"""
    import subprocess
    from test_host import _build_cpp_shim
    r = subprocess.run([_build_cpp_shim(built_library), "gpu"], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


@pytest.mark.parametrize("N,K,L", [(256, 128, 3), (256, 96, 5), (512, 256, 6), (128, 64, 12), (1024, 512, 7),
                                    (8192, 4096, 1), (8192, 4096, 4), (4096, 1024, 8), (16384, 8192, 1)])
def test_odd_list_sizes_and_large_blocks(built_library, N, K, L):
    """list sizes that are not powers of two (the reference accepts any L) and block lengths above the
    BASELINE configs, on Bhattacharyya frozen sets built by pypolar_amd.construction (= the reference's)."""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    rng = np.random.default_rng(N + 17 * L)
    frozen = frozen_bits(N, K, 0.0, "BB")
    det = 8
    B = 24 if N <= 1024 else 6
    rows = []
    for i in range(B):
        info = rng.integers(0, 256, K // 8, dtype=np.uint8)
        _, cw = ob.encode(N, frozen, det, info)
        x = 1.0 - 2.0 * np.unpackbits(cw)[:N]
        rows.append(x + rng.normal(0, 0.95, N) if i % 2 else np.round((x + rng.normal(0, 0.9, N)) * 4))
    llr = np.asarray(rows, np.float32)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    res = d.decode_batch(llr, return_metrics=L > 1)
    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr, True, "float")
    assert np.array_equal(res[0], e_out)
    assert np.array_equal(res[1].astype(np.uint8), e_ok)
    if L > 1:
        assert np.allclose(res[2], e_met, rtol=METRIC_RTOL, atol=0)


HANDMADE = [
    # frozen sets no construction produces but the API accepts; they reach the SC nodes the BB/BE sets never
    # use (SURVEY Q8 / App. A.1 rule 11a: ZEROSPC needs index 7 frozen while 9 is information; plain n = 8
    # blocks without a special decoder; DSPC with n > 8; TREP)
    (16, [0, 1, 2, 3, 4, 5, 6, 7, 8]),                 # left half frozen, right half one frozen -> ZEROSPC
    (32, list(range(16)) + [16]),                      # ZEROSPC at n = 32
    (32, [0, 1]),                                      # DSPC, n = 32
    (16, list(range(13))),                             # TREP  (c = n-3, last frozen = n-4)
    (8, [0, 1, 4]),                                    # n = 8 without a special node -> ShortRateR
    (16, [0, 1, 2, 4, 8]),                             # generic split, RateR + small leaves
    (64, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 17, 18, 20, 24, 32, 33, 34, 36, 40, 48]),
]


@pytest.mark.parametrize("N,frozen", HANDMADE, ids=[f"N{n}_F{len(f)}" for n, f in HANDMADE])
@pytest.mark.parametrize("L", [1, 4, 8])
def test_handmade_frozen_sets(built_library, N, frozen, L):
    import pypolar_amd
    rng = np.random.default_rng(N * 7 + len(frozen) + L)
    F = np.asarray(frozen, np.uint32)
    llr = np.concatenate([rng.normal(0, 1, (96, N)), rng.choice([-1.0, 1.0, 2.0, -2.0], (96, N))]).astype(np.float32)
    d = pypolar_amd.PolarDecoder(N, L, F, "float")
    d.setErrorDetection(0, "crc")
    res = d.decode_batch(llr, return_metrics=L > 1)
    e_out, e_ok, e_met = ob.decode_batch(N, L, F, 0, llr, True, "float")
    assert np.array_equal(res[0], e_out)
    if L > 1:
        assert np.allclose(res[2], e_met, rtol=METRIC_RTOL, atol=0)


@pytest.mark.parametrize("name,reps", [("cfg3_n1024_k512_l8_crc24_awgn.bin", 400), ("mix_n1024_k512_l8_crc24_awgn.bin", 900),
                                       ("cfg1_n64_k40_sc_awgn.bin", 7000)])
def test_large_host_pointer_batches(built_library, name, reps):
    """host (numpy) batches of 100+ MB through the staging path of pcd_decode_f32 (ragged row count): outputs,
    verdicts and metrics must come out in place"""
    fx = gio.load_named(name)
    d = _decoder(fx)
    llr = np.tile(fx.llr, (reps, 1))[: fx.count * reps - 37]            # ragged tail
    want_m = fx.L > 1 and fx.dectype == "float"
    res = d.decode_batch(llr, return_metrics=want_m)
    exp_out = np.tile(fx.out, (reps, 1))[: llr.shape[0]]
    exp_ok = np.tile(fx.ok, reps)[: llr.shape[0]]
    assert np.array_equal(res[0], exp_out)
    assert np.array_equal(res[1].astype(np.uint8), exp_ok)
    if want_m:
        exp_m = np.tile(fx.metrics, (reps, 1))[: llr.shape[0]]
        assert np.array_equal(res[2].view(np.uint32), exp_m.view(np.uint32))


@pytest.mark.parametrize("N", [16, 32, 64, 128, 256, 512])
def test_random_frozen_sets_all_list_sizes(built_library, N):
    """frozen sets no construction would produce (arbitrary subsets): every tree shape the node-selection rules can
    emit, incl. the sets for which the reference's SC builder throws (both sides must refuse the same sets)"""
    import pypolar_amd
    rng = np.random.default_rng(7000 + N)
    refused = 0
    for trial in range(10):
        K = int(rng.integers(1, N))
        if trial == 0:
            K = N - 1
        if trial == 1:
            K = 1
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
        nb = (K + 7) // 8
        det = 8 if (nb >= 2 and K % 8 == 0) else 0
        B = 20
        llr = np.concatenate([rng.normal(0.3, 1.0, (B // 2, N)), rng.choice([-2.0, -1.0, 1.0, 2.0], (B // 2, N))]).astype(np.float32)
        for L in (1, 2, 4, 8, 16):
            try:
                e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, llr)
            except ValueError:
                assert L == 1
                with pytest.raises(ValueError):
                    pypolar_amd.PolarDecoder(N, L, frozen, "float")
                refused += 1
                continue
            d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
            d.setErrorDetection(det, "crc")
            if L == 1:
                out, ok = d.decode_batch(llr)
            else:
                out, ok, met = d.decode_batch(llr, return_metrics=True)
                assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), (N, K, L, trial)
            assert np.array_equal(out, e_out), (N, K, L, trial)
            assert np.array_equal(ok.astype(np.uint8), e_ok), (N, K, L, trial)
    print(f"N={N}: {refused} sets refused by both SC builders")


@pytest.mark.parametrize("L", [1, 2, 4, 8, 16])
def test_maximum_block_length(built_library, L):
    """N = 32768 (the largest block length the ABI accepts): SC and list decoding against the oracle"""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    N, K = 32768, 16384
    frozen = frozen_bits(N, K, 0.0, "BB")
    rng = np.random.default_rng(50 + L)
    info = rng.integers(0, 256, (6, K // 8), dtype=np.uint8)
    llr = np.empty((6, N), np.float32)
    for r in range(6):
        _, cw = ob.encode(N, frozen, 32, info[r])
        llr[r] = 1.0 - 2.0 * np.unpackbits(cw)[:N] + rng.normal(0, 0.75 if r < 4 else 1.5, N)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    d.setErrorDetection(32, "crc")
    e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 32, llr)
    if L == 1:
        out, ok = d.decode_batch(llr)
    else:
        out, ok, met = d.decode_batch(llr, return_metrics=True)
        assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32))
    assert np.array_equal(out, e_out) and np.array_equal(ok.astype(np.uint8), e_ok)
    assert e_ok[:4].all()


@pytest.mark.parametrize("N", [8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_grid_of_block_lengths_list_sizes_and_rates(built_library, N):
    """every kernel instantiation and geometry once: all list sizes (incl. the ones between powers of two) x three
    rates per block length, float / int8 input, systematic or not, a few noisy and tie-heavy rows each"""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    rng = np.random.default_rng(900 + N)
    logN = N.bit_length() - 1
    combos = 0
    for L in (1, 2, 3, 4, 5, 7, 8, 12, 16, 24, 32):
        lp = 1
        while lp < L:
            lp <<= 1
        if L > 1 and logN * (lp.bit_length() - 1) > 64:
            with pytest.raises(pypolar_amd.PolarAmdError):      # PCD_ERR_CAPACITY: documented limit
                pypolar_amd.PolarDecoder(N, L, frozen_bits(N, N // 2, 0.0, "BB"), "float")
            continue
        for R in (1, 2, 3):
            K = max(1, (N * R // 4) // 8 * 8) if N >= 32 else max(1, N * R // 4)
            if K >= N:
                continue
            frozen = frozen_bits(N, K, 0.0, "BE" if (L + R) % 2 else "BB")
            nb = (K + 7) // 8
            det = 8 if (K % 8 == 0 and nb >= 2) else 0
            variant = (L + R + logN) % 3            # 0: float, 1: non-systematic, 2: int8 input
            B = 4 if N >= 8192 else 6
            llr = np.concatenate([rng.normal(0.4, 1.0, (B // 2, N)) * 8.0, rng.choice([-3.0, -1.0, 1.0, 3.0], (B - B // 2, N))])
            if variant == 2:
                x = np.clip(np.round(llr), -127, 127).astype(np.int8)
                ref_in = x.astype(np.float32)
            else:
                x = llr.astype(np.float32)
                ref_in = x
            d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
            d.setErrorDetection(det, "crc")
            d.setSystematic(variant != 1)
            e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, ref_in, systematic=(variant != 1))
            if L == 1:
                out, ok = d.decode_batch(x)
            else:
                out, ok, met = d.decode_batch(x, return_metrics=True)
                assert np.array_equal(met.view(np.uint32), e_met.view(np.uint32)), (N, K, L, variant)
            assert np.array_equal(out, e_out), (N, K, L, variant)
            assert np.array_equal(ok.astype(np.uint8), e_ok), (N, K, L, variant)
            combos += 1
    assert combos >= 20


@pytest.mark.parametrize("N,L", [(64, 2), (256, 4), (1024, 8), (1024, 32), (4096, 8), (4096, 16),
                                 (128, 8), (512, 2), (512, 4), (512, 16), (1024, 4), (2048, 2), (2048, 4)])   # (round 4: the pair kernels below N = 1024 in front of the 2 / 4-lane list kernels)
def test_mixed_decoder_grid(built_library, N, L):
    """AdaptiveFloat (SC first, list decoding for the rows the detector rejects) against the oracle at several sizes:
    rows near the SC failure threshold so that both branches are taken"""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    K = N // 2
    frozen = frozen_bits(N, K, 0.0, "BB")
    rng = np.random.default_rng(31 * N + L)
    B = 48
    info = rng.integers(0, 256, (B, K // 8), dtype=np.uint8)
    llr = np.empty((B, N), np.float32)
    for r in range(B):
        _, cw = ob.encode(N, frozen, 8, info[r])
        llr[r] = 1.0 - 2.0 * np.unpackbits(cw)[:N] + rng.normal(0, 0.55 + 0.5 * (r % 4) / 3.0, N)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "mixed")
    d.setErrorDetection(8, "crc")
    out, ok = d.decode_batch(llr)
    e_out, e_ok, _ = ob.decode_batch(N, L, frozen, 8, llr, dectype="mixed")
    assert np.array_equal(out, e_out) and np.array_equal(ok.astype(np.uint8), e_ok)
    sc_out, sc_ok, _ = ob.decode_batch(N, 1, frozen, 8, llr)
    assert 0 < int(sc_ok.sum()) < B      # the batch exercises both the SC-only and the SC + list branch


FUSE_SWITCHES = ["PCD_NO_FUSE", "PCD_NO_FUSE_C", "PCD_NO_FUSE_R0", "PCD_NO_FUSE_G0", "PCD_NO_FUSE_GRAND", "PCD_NO_FUSE_CW",
                 "PCD_NO_FUSE_F", "PCD_NO_FUSE_G", "PCD_SC_NO_SKEW", "PCD_NO_VTOP", "PCD_NO_VIRTUAL", "PCD_NO_WFG", "PCD_VG2", "PCD_NO_LANE", "PCD_FORCE_LANE", "PCD_NO_PAIR"]


@pytest.mark.parametrize("switch", FUSE_SWITCHES)
def test_fused_and_unfused_programs_agree(built_library, switch):
    """The host rewrites the decoders' instruction lists (capi.cpp: specialize_scl -- leaf prologues, COMBINE chains,
    G0_S, SCL_VTOP; fuse_sc -- leaf prologues / epilogues, COMBINE chains, tail G, left-descent F runs); every stage of
    that rewrite can be switched off in the DIAGNOSTIC build of the library (`make diag`: the release library reads no
    environment variable).  All variants must reproduce the reference's vectors -- SC, 4 lanes per path (L = 8),
    8 lanes per path (L = 2, 4) and random frozen sets.  One child process per switch (tests/diag_fuse_check.py) loads
    the diagnostic library; the release library of this process is untouched."""
    import subprocess
    import sys
    import pypolar_amd
    diag = os.path.join(os.path.dirname(pypolar_amd.library_path()), "libpolar_amd_diag.so")
    if not os.path.exists(diag):
        subprocess.check_call(["make", "-C", os.path.dirname(os.path.dirname(diag)), "diag"])
    env = dict(os.environ, PYPOLAR_AMD_LIB=diag)
    env[switch] = "1"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag_fuse_check.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (switch, r.stdout[-2000:], r.stderr[-2000:])


def test_release_library_reads_no_environment(built_library, monkeypatch):
    """a stray experiment variable must not change (let alone break) the shipped decoder"""
    monkeypatch.setenv("PCD_ABLATE", "255")
    monkeypatch.setenv("PCD_NO_FUSE", "1")
    monkeypatch.setenv("PCD_S0", "3")
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    res = _decoder(fx).decode_batch(fx.llr[:16])
    assert np.array_equal(res[0], fx.out[:16])
    import pypolar_amd
    with open(pypolar_amd.library_path(), "rb") as f:
        assert b"PCD_" not in f.read()




# ---- int8 ("char") decoders: FastSscFipChar / SclFipChar -------------------------------------------------------
def test_char_round_trip_like_polarcodetest(built_library):
    """test/polarcode/polarcodetest.cpp:45-183: N = 4096, K = 3072 (Bhattacharyya), systematic encode, BPSK to +-1
    int8, FastSscFipChar and SclFipChar (L = 4, CRC-32) must return the information word."""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    N, K = 4096, 3072
    frozen = frozen_bits(N, K, 0.0, "BB")
    rng = np.random.default_rng(5489)
    for L, crc in ((1, 8), (4, 32)):
        info = rng.integers(0, 256, (16, K // 8), dtype=np.uint8)
        enc = pypolar_amd.PolarEncoder(N, frozen)
        enc.setErrorDetection(crc, "crc")
        code = enc.encode_batch(info)                      # (fills the checksum field of `info`)
        sig = (1 - 2 * np.unpackbits(code, axis=1).astype(np.int16)).astype(np.int8)   # bpskModulate :27-41
        d = pypolar_amd.PolarDecoder(N, L, frozen, "char")
        d.setErrorDetection(crc, "crc")
        out, ok = d.decode_batch(sig)
        assert ok.all() and np.array_equal(out, info)
        assert np.array_equal(d.decode_vector(sig[0]), info[0])


@pytest.mark.parametrize("N,L", [(64, 1), (64, 4), (256, 8), (1024, 1), (1024, 8), (2048, 2), (512, 16), (4096, 1)])
def test_char_decoders_vs_oracle_random(built_library, N, L):
    """seeded int8 rows (full range incl. -128 and 0, clipped Gaussians, +-1) on Bhattacharyya and random frozen sets"""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    rng = np.random.default_rng(N * 131 + L)
    sets = [frozen_bits(N, N // 2, 0.0, "BB"), frozen_bits(N, (N // 4) // 8 * 8 or 8, 0.0, "BE"),
            np.sort(rng.choice(N, N // 2, replace=False)).astype(np.uint32)]
    for frozen in sets:
        K = N - len(frozen)
        det = 8 if K >= 16 and K % 8 == 0 else 0
        rows = np.concatenate([
            rng.integers(-128, 128, (24, N)),
            np.clip(np.rint(rng.normal(6.0, 9.0, (24, N))), -128, 127),
            np.clip(np.rint(rng.normal(40.0, 60.0, (24, N))), -128, 127),
            rng.choice([-1, 0, 1], (16, N)),
        ]).astype(np.int8)
        d = pypolar_amd.PolarDecoder(N, L, frozen, "char")
        d.setErrorDetection(det, "crc")
        res = d.decode_batch(rows, return_metrics=L > 1)
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, det, rows.astype(np.float32), True, "char")
        assert np.array_equal(res[0], e_out), (N, L, K)
        assert np.array_equal(res[1].astype(np.uint8), e_ok), (N, L, K)
        if L > 1:
            assert np.array_equal(res[2], e_met), (N, L, K)      # exact integers
        d.setSystematic(False)
        res = d.decode_batch(rows[:32])
        e_out, e_ok, _ = ob.decode_batch(N, L, frozen, det, rows[:32].astype(np.float32), False, "char")
        assert np.array_equal(res[0], e_out) and np.array_equal(res[1].astype(np.uint8), e_ok), (N, L, K, "nsys")


@pytest.mark.parametrize("N,L", [(256, 4), (1024, 8), (2048, 8), (512, 16), (256, 32), (1024, 2)])
def test_char_list_decoder_equal_metrics_everywhere(built_library, N, L):
    """int8 rows of tiny magnitude: nearly every candidate selection has equal metrics among the winners and across the
    cut, so the order simplePartialSortDescending (arrayfuncs.h:161-183) leaves them in -- including the swaps that move
    candidates around -- decides which paths survive and where (scl_kernel.h, select_literal_cw): bytes, verdicts and all
    L metrics must equal the oracle's.  tools/char_stress_vs_oracle.py is the long form."""
    import pypolar_amd
    from pypolar_amd.construction import frozen_bits
    rng = np.random.default_rng(N + L)
    for rate, amp in ((0.25, 1), (0.5, 1), (0.5, 3), (0.75, 2), (0.5, 12)):
        K = int(N * rate) // 8 * 8
        frozen = np.asarray(frozen_bits(N, K, 0.0, "BB"), dtype=np.uint32)
        rows = rng.integers(-amp, amp + 1, (96, N)).astype(np.int8)
        d = pypolar_amd.PolarDecoder(N, L, frozen, "char")
        d.setErrorDetection(8, "crc")
        out, ok, met = d.decode_batch(rows, return_metrics=True)
        e_out, e_ok, e_met = ob.decode_batch(N, L, frozen, 8, rows.astype(np.float32), True, "char")
        assert np.array_equal(out, e_out), (N, L, K, amp)
        assert np.array_equal(np.asarray(ok, np.uint8), e_ok), (N, L, K, amp)
        assert np.array_equal(met, e_met), (N, L, K, amp)


def test_char_decoder_float_rows_are_converted_like_the_reference(built_library):
    """CharContainer::insertLlr(const float*) bitcontainer.cpp:505-516 (N >= 8): round to nearest EVEN, saturate"""
    import pypolar_amd
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    x = (fx.llr[:32] * 9.5).astype(np.float32)
    x[0, :8] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 300.0, -300.0]
    conv = np.clip(np.rint(x), -128, 127).astype(np.int8)        # np.rint rounds half to even
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, "char")
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    a = d.decode_batch(x)
    b = d.decode_batch(conv)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    e_out, e_ok, _ = ob.decode_batch(fx.N, fx.L, fx.frozen, fx.crc_size, conv.astype(np.float32), True, "char")
    assert np.array_equal(a[0], e_out) and np.array_equal(a[1].astype(np.uint8), e_ok)


def test_large_host_batch_goes_through_the_pinned_ring(built_library):
    """host pointers, >= 96 MiB of rows: pinned-ring pipeline (capi.cpp decode_host_pipelined) -- ragged last chunk,
    fp32 and int8 rows, metrics; must equal the small-batch path row for row"""
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    d = _decoder(fx)
    reps = 40000 // fx.count + 1
    rows = np.tile(fx.llr, (reps, 1))[:40003]                       # 156 MiB of fp32 rows, not a multiple of the chunk
    out, ok, met = d.decode_batch(rows, return_metrics=True)
    idx = np.arange(rows.shape[0]) % fx.count
    assert np.array_equal(out, fx.out[idx]) and np.array_equal(ok.astype(np.uint8), fx.ok[idx])
    assert np.array_equal(met.view(np.uint32), fx.metrics[idx].view(np.uint32))
    fq = gio.load_named("i8_cfg3_n1024_k512_l8_crc24.bin")
    dq = _decoder(fq)
    reps = 120000 // fq.count + 1
    rows8 = np.tile(fq.llr.astype(np.int8), (reps, 1))[:120001]     # 117 MiB of int8 rows
    out, ok, met = dq.decode_batch(rows8, return_metrics=True)      # (int8 rows: the results are streamed through the pinned staging buffer, metrics included)
    idx = np.arange(rows8.shape[0]) % fq.count
    assert np.array_equal(out, fq.out[idx]) and np.array_equal(ok.astype(np.uint8), fq.ok[idx])
    assert np.array_equal(met.view(np.uint32), fq.metrics[idx].view(np.uint32))


@pytest.mark.parametrize("N,L", [(64, 1), (128, 1), (256, 1), (128, 8), (256, 2), (1024, 4), (2048, 3), (64, 8)])
def test_decoders_whose_lone_wave_is_quick_have_no_sibling(built_library, N, L):
    """SC at N <= 256 (sc_pair_n7 / n8_kernel) and the list decoders on the 2 / 4-lane kernels or at N <= 256: a lone wave of the
    throughput kernel answers a call of one row as fast as a one-codeword-per-wave sibling would (tools/latency_small_n.py), so there is
    none -- pcd_small_batch_rows() is 0 -- and one row, a few rows and a ragged wave decode to the oracle's answers"""
    import pypolar_amd
    pypolar_amd.SMALL_BATCH_ROWS = None          # (the library's own routing)
    frozen = np.asarray(pypolar_amd.frozen_bits(N, N // 2, 0.0, "BB"), dtype=np.uint32)
    rng = np.random.default_rng(N + L)
    llr = ((1.0 + 0.8 * rng.standard_normal((41, N))) * np.where(rng.random((41, N)) < 0.5, 1, -1)).astype(np.float32)
    d = pypolar_amd.PolarDecoder(N, L, frozen, "float")
    d.setErrorDetection(8, "crc")
    assert d.small_batch_rows() == 0
    e_out, e_ok, _ = ob.decode_batch(N, L, frozen, 8, llr, True, "float")
    for rows in (1, 3, 33, 41):
        out, ok = d.decode_batch(llr[:rows])
        assert np.array_equal(out, e_out[:rows]) and np.array_equal(ok.astype(np.uint8), e_ok[:rows])


@pytest.mark.parametrize("name", ["cfg2_n1024_k512_sc_awgn.bin", "cfg3_n1024_k512_l8_crc24_awgn.bin", "cfg4_n1024_k176_l8_crc11_awgn.bin",
                                  "mix_n1024_k512_l8_crc24_awgn.bin", "char_cfg3_n1024_k512_l8_crc24.bin", "cfg5_n4096_k2048_l32_crc32_awgn.bin"])
def test_small_batches_take_the_low_latency_geometry_with_identical_results(built_library, name):
    """pcd_small_batch_rows: calls of one .. a few hundred rows (the reference's interface is one codeword per call,
    decoder_python.cc:41-74) run on a sibling handle with one codeword per wavefront; bytes, verdicts and metrics are the
    reference's either way, for host and device pointers, at the threshold and one row above it"""
    import pypolar_amd
    import torch
    fx = gio.load(os.path.join(gio.GOLDEN_DIR, name))[0]
    pypolar_amd.SMALL_BATCH_ROWS = None          # (the library's own routing)
    d = pypolar_amd.PolarDecoder(fx.N, fx.L, fx.frozen, fx.dectype)
    d.setErrorDetection(fx.crc_size, fx.crc_type)
    thr = d.small_batch_rows()
    assert thr >= 256, thr                        # (these decoders run on wave-packed kernels: they have a sibling)
    x = fx.llr.astype(np.int8) if fx.dectype == "char" else fx.llr
    for rows in (1, 3, 17):
        out, ok, met = d.decode_batch(np.ascontiguousarray(x[:rows]), return_metrics=True)
        assert np.array_equal(out, fx.out[:rows]) and np.array_equal(ok.astype(np.uint8), fx.ok[:rows])
        if fx.L > 1 and fx.dectype != "mixed":
            assert np.array_equal(met.view(np.uint32), fx.metrics[:rows].view(np.uint32))
    # across the threshold, device pointers: thr rows (sibling) and thr + 1 rows (throughput kernels) agree row by row
    d.set_small_batch_rows(40)
    xd = torch.from_numpy(np.ascontiguousarray(x[:41])).cuda()
    a = d.decode_batch(xd[:40])
    b = d.decode_batch(xd)
    torch.cuda.synchronize()
    assert bool((a[0] == b[0][:40]).all()) and bool((a[1] == b[1][:40]).all())
    assert np.array_equal(b[0].cpu().numpy(), fx.out[:41])
    d.set_small_batch_rows(0)
    c = d.decode_batch(xd[:5])
    torch.cuda.synchronize()
    assert np.array_equal(c[0].cpu().numpy(), fx.out[:5])
