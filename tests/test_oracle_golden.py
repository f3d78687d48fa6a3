"""CPU suite, part 1: pin the oracle (oracle/polar_oracle.c) against
  * every golden vector produced by the unmodified reference library (tests/golden/*.bin), and
  * the reference's own known-answer tests (decodingtest.cpp, qa_pypolar_detector.py).
Nothing here touches the product or a GPU."""
import numpy as np
import pytest

import golden_io as gio
import oracle_binding as ob

FILES = gio.all_files()


def test_fixture_inventory():
    names = [f.split("/")[-1] for f in FILES]
    for cfg in ("cfg1_", "cfg2_", "cfg3_", "cfg4_", "cfg5_", "mix_", "sweep_", "nsys_cfg2", "nsys_cfg3", "nsys_cfg4",
                "nsys_sweep", "nsys_mix", "i8_cfg2", "i8_cfg3", "i8_cfg4", "i8_sweep", "i8_nsys",
                "char_cfg2", "char_cfg3", "char_cfg4", "char_sweep", "char_n4096", "char_nsys"):
        assert any(n.startswith(cfg) for n in names), cfg


@pytest.mark.parametrize("path", FILES, ids=[f.split("/")[-1] for f in FILES])
def test_oracle_matches_reference_vectors(path):
    for fx in gio.load(path):
        # ("char" fixtures: FastSscFipChar / SclFipChar restated in the same oracle, rows are int8-valued floats)
        # (int8 input: FloatContainer::insertLlr(const char*) is an exact widening, the fixture keeps the widened row)
        out, ok, met = ob.decode_batch(fx.N, fx.L, fx.frozen, fx.crc_size, fx.llr, fx.systematic, fx.dectype)
        assert np.array_equal(out, fx.out), fx.name
        assert np.array_equal(ok, fx.ok), fx.name
        if fx.L > 1 and fx.dectype in ("float", "char"):
            for r in range(fx.count):
                p = int(fx.npaths[r])
                # bit-identical, i.e. far inside the 1e-5 relative tolerance of the north star
                assert np.array_equal(met[r, :p].view(np.uint32), fx.metrics[r, :p].view(np.uint32)), (fx.name, r)


# ---- f / g / combine known answers: test/polarcode/decodingtest.cpp:471-503 ---------------------
def test_f_g_known_answers():
    import ctypes as C
    lib = ob.lib()
    a = np.array([-7, 6, 5, 4, 3, 2, 1, 0], np.float32)
    b = np.array([8, -7, 6, 5, -4, 3, 2, 1], np.float32)
    x = np.concatenate([a, b]).astype(np.float32)
    out = np.zeros(8, np.float32)
    lib.po_f(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 8)
    assert out.tolist() == [-7, -6, 5, 4, -3, 2, 1, 0]
    b2 = np.array([8, -7, 6, 5, -4, 3, 2, -1], np.float32)
    bits = np.array([-128, -128, 0, 0, -128, 0, 0, 0], np.float32)
    x2 = np.concatenate([a, b2]).astype(np.float32)
    lib.po_g(x2.ctypes.data_as(C.c_void_p), bits.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 8)
    assert out.tolist() == [15, -13, 11, 9, -7, 5, 3, -1]


# ---- N = 8 known answer: decodingtest.cpp:1031-1059,1137-1180 ----------------------------------
@pytest.mark.parametrize("L", [1, 4])
def test_n8_known_answer(L):
    frozen = np.array([0, 1, 2, 4], np.uint32)
    llr = np.array([[-5, -6, -4, 1, -4, -5, -7, 2]], np.float32)
    out, ok, _ = ob.decode_batch(8, L, frozen, 0, llr)
    assert (out[0, 0] & 0xF0) == 0xF0


# ---- closed-form leaf checks: decodingtest.cpp:159-282 -------------------------------------------
@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 256, 512])
def test_repetition_leaf_iota(n):
    # a length-n code with only the last bit unfrozen is one REP leaf; LLR = iota => positive sum => bit 0
    frozen = np.arange(n - 1, dtype=np.uint32)
    llr = np.arange(n, dtype=np.float32).reshape(1, n)
    out, _, _ = ob.decode_batch(n, 1, frozen, 0, llr)
    assert out[0, 0] == 0
    out, _, _ = ob.decode_batch(n, 1, frozen, 0, -llr)
    assert out[0, 0] == 0x80


@pytest.mark.parametrize("n", [8, 16, 32, 64, 128, 256, 512])
def test_spc_leaf_flips_least_reliable(n):
    # one frozen bit at index 0 => a single SPC leaf; decodingtest.cpp:245-282: iota(-2.9...) has odd
    # parity (3 negatives) and the smallest magnitude at index 3, so only element 3 flips
    frozen = np.array([0], np.uint32)
    llr = (np.arange(n, dtype=np.float32) - 2.9).reshape(1, n)
    import ctypes as C
    bits = np.zeros(n, np.uint8)
    rc = ob.lib().po_sc_decode_bits(n, frozen.ctypes.data_as(C.c_void_p), 1, llr.ctypes.data_as(C.c_void_p),
                                    bits.ctypes.data_as(C.c_void_p))
    assert rc == 0
    expect = (llr[0] < 0).astype(np.uint8)
    expect[3] ^= 1
    assert np.array_equal(bits, expect)


# ---- CRC known answers: python/qa_pypolar_detector.py:82-105,269-283,383-402 ---------------------
def test_crc_known_answers():
    msg = np.frombuffer(b"123456789", np.uint8)
    assert ob.det_calc(24, msg) == 0xF48279          # CRC-24C
    assert ob.det_calc(11, msg) == 0x5CA             # CRC-11 (stored as 05 ca)
    assert ob.det_calc(32, np.frombuffer(b"Test", np.uint8)) == int.from_bytes(bytes([0x8C, 0x2D, 0xE2, 0x19]), "little")
    g = ob.det_generate(24, np.concatenate([msg, np.zeros(3, np.uint8)]))
    assert bytes(g[-3:]) == bytes([0xF4, 0x82, 0x79]) and ob.det_check(24, g)
    g = ob.det_generate(11, np.concatenate([msg, np.zeros(2, np.uint8)]))
    assert bytes(g[-2:]) == bytes([0x05, 0xCA]) and ob.det_check(11, g)
    g[0] ^= 1
    assert not ob.det_check(11, g)


# ---- generate -> check -> corrupt -> fail: test/polarcode/errordetectiontest.cpp:47-73 -----------
@pytest.mark.parametrize("det", [8, 11, 24, 32])
def test_detector_roundtrip(det):
    rng = np.random.default_rng(det)
    for nbytes in (8, 16, 64):
        data = rng.integers(0, 256, nbytes, dtype=np.uint8)
        g = ob.det_generate(det, data)
        assert ob.det_check(det, g)
        g[1] ^= 0x10
        assert not ob.det_check(det, g)


# ---- encode -> decode round trip of python/qa_pypolar_decoder.py:94-132 on the oracle -------------
@pytest.mark.parametrize("N", [128, 256, 512, 1024])
@pytest.mark.parametrize("L", [1, 4, 8])
def test_roundtrip_like_qa_pypolar(N, L):
    rng = np.random.default_rng(N * 31 + L)
    fx = {128: "sweep_n128.bin", 256: "sweep_n256.bin", 512: "sweep_n512.bin", 1024: "cfg3_n1024_k512_l8_crc24_awgn.bin"}[N]
    frozen = gio.load(gio.GOLDEN_DIR + "/" + fx)[0].frozen
    K = N - frozen.size
    for _ in range(4):
        info = rng.integers(0, 256, K // 8, dtype=np.uint8)
        info_crc, cw = ob.encode(N, frozen, 8, info)
        bits = np.unpackbits(cw)
        llr = (1.0 - 2.0 * bits + rng.uniform(-0.2, 0.2, N)).astype(np.float32).reshape(1, N)
        out, ok, _ = ob.decode_batch(N, L, frozen, 8, llr)
        assert ok[0] == 1 and np.array_equal(out[0], info_crc)


# ---- encoder: oracle po_encode vs the reference's ButterflyFipPacked::encode_vector (tests/golden/enc.bin) ----
def test_oracle_encoder_matches_reference_vectors():
    fxs = gio.load_enc()
    assert len(fxs) > 60
    rows = 0
    for fx in fxs:
        for r in range(fx.count):
            info, cw = ob.encode(fx.N, fx.frozen, fx.crc_size, fx.info_in[r], fx.systematic)
            assert np.array_equal(info, fx.info_after[r]), (fx.name, r)
            assert np.array_equal(cw, fx.cw[r]), (fx.name, r, fx.N, fx.K, fx.systematic)
            rows += 1
    assert rows > 400


# ---- puncturer: oracle vs the reference's Puncturer (tests/golden/punc.bin) -------------------------------------
def test_oracle_puncturer_matches_reference_vectors():
    fxs = gio.load_punc()
    assert len(fxs) >= 15
    for fx in fxs:
        pos, parent = ob.punc_positions(fx.M, fx.frozen)
        assert parent == fx.N and np.array_equal(pos, fx.positions), fx.name
        for r in range(fx.count):
            assert np.array_equal(ob.puncture_f32(pos, fx.x[r]), fx.punctured[r]), fx.name
            assert np.array_equal(ob.depuncture_f32(pos, fx.N, fx.punctured[r]), fx.depunctured[r]), fx.name
            if fx.M % 8 == 0:
                assert np.array_equal(ob.puncture_packed(pos, fx.packed_in[r]), fx.packed_out[r]), fx.name
    with pytest.raises(IndexError):   # std::out_of_range puncturer.cpp:56-59
        ob.punc_positions(513, np.arange(10, dtype=np.uint32))
