"""GPU suite (-m gpu), puncturer row: the HIP puncture / depuncture kernels through the C ABI
(pypolar_amd.Puncturer -> ppu_*) against golden vectors of the unmodified reference's PolarCode::Puncturer
(tests/golden/punc.bin), numpy indexing on random data for every element type, and a punctured end-to-end chain
encode -> puncture -> channel -> depuncture -> decode at a large batch.  Bar: exact equality (pure data movement)."""
import numpy as np
import pytest

import golden_io as gio

pytestmark = pytest.mark.gpu


def test_hip_puncturer_matches_reference_vectors(built_library):
    import pypolar_amd
    for fx in gio.load_punc():
        p = pypolar_amd.Puncturer(fx.M, fx.frozen)
        assert np.array_equal(p.puncture_batch(fx.x), fx.punctured), fx.name
        assert np.array_equal(p.depuncture_batch(fx.punctured), fx.depunctured), fx.name
        if fx.M % 8 == 0:
            assert np.array_equal(p.puncture_packed_batch(fx.packed_in), fx.packed_out), fx.name
        # the reference's one-vector interface and its dtype overloads (double / float / uint8)
        r = p.puncture(fx.x[0])
        assert r.dtype == np.float32 and np.array_equal(r, fx.punctured[0])
        r = p.puncture(fx.x[0].astype(np.float64))
        assert r.dtype == np.float64 and np.array_equal(r, fx.punctured[0].astype(np.float64))
        r = p.depuncture(fx.punctured[0])
        assert np.array_equal(r, fx.depunctured[0])
        bits = np.unpackbits(fx.packed_in[0])
        r = p.puncture(bits)
        assert r.dtype == np.uint8 and np.array_equal(r, bits[fx.positions])
        if fx.M % 8 == 0:   # qa_pypolar_puncturer.py:61-66
            assert np.array_equal(np.unpackbits(p.puncturePacked(fx.packed_in[0])), r)
        r = p.puncture(np.arange(fx.N, dtype=np.int32))      # any other dtype converts to the double overload
        assert r.dtype == np.float64 and np.array_equal(r, fx.positions.astype(np.float64))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8])
def test_random_rows_all_element_types(built_library, dtype):
    import pypolar_amd
    rng = np.random.default_rng(4)
    for fx in gio.load_punc()[::3]:
        p = pypolar_amd.Puncturer(fx.M, fx.frozen)
        for B in (0, 1, 37, 300):
            x = (rng.normal(0, 50, (B, fx.N))).astype(dtype) if dtype != np.uint8 else rng.integers(0, 256, (B, fx.N), dtype=np.uint8)
            pun = p.puncture_batch(x)
            assert pun.dtype == dtype and np.array_equal(pun, x[:, fx.positions])
            dep = p.depuncture_batch(pun)
            ref = np.zeros((B, fx.N), dtype)
            ref[:, fx.positions] = pun
            assert np.array_equal(dep, ref)


def test_punctured_chain_at_full_batch(built_library):
    """rate matching around the codec, device resident, 2^19 rows: encode (N = 1024) -> puncturePacked to 864 bits ->
    BPSK -> depuncture (zeros = erasures on the removed frozen positions) -> list decoder: every payload comes back"""
    import torch
    import pypolar_amd
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    N, K, nb, M = fx.N, fx.K, (fx.K + 7) // 8, 864
    B = 1 << 19
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    info = torch.randint(0, 256, (B, nb), dtype=torch.uint8, device=dev, generator=g)
    enc = pypolar_amd.PolarEncoder(N, fx.frozen)
    enc.setErrorDetection(24, "crcnr")
    code = enc.encode_batch(info)
    punc = pypolar_amd.Puncturer(M, fx.frozen)
    pos = torch.from_numpy(np.asarray(punc.blockOutputPositions(), np.int64)).to(dev)
    sent = punc.puncture_packed_batch(code)
    assert sent.shape == (B, M // 8)
    shifts = torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)
    cbits = ((code.unsqueeze(-1) >> shifts) & 1).reshape(B, N)
    sbits = ((sent.unsqueeze(-1) >> shifts) & 1).reshape(B, M)
    assert torch.equal(sbits, cbits[:, pos])
    rx = 1.0 - 2.0 * sbits.float()
    llr = punc.depuncture_batch(rx)
    assert llr.shape == (B, N)
    ref = torch.zeros((B, N), dtype=torch.float32, device=dev)
    ref[:, pos] = rx
    assert torch.equal(llr, ref)
    dec = pypolar_amd.PolarDecoder(N, 8, fx.frozen, "float")
    dec.setErrorDetection(24, "crcnr")
    out, ok = dec.decode_batch(llr)
    assert bool(ok.all()) and torch.equal(out, info)
    # double precision and byte elements on device pointers
    xd = torch.randn((4096, N), dtype=torch.float64, device=dev, generator=g)
    assert torch.equal(punc.puncture_batch(xd), xd[:, pos])
    xb = torch.randint(0, 256, (4096, M), dtype=torch.uint8, device=dev, generator=g)
    db = punc.depuncture_batch(xb)
    assert torch.equal(db[:, pos], xb) and int(db.sum()) == int(xb.sum())
