"""GPU suite (-m gpu), encoder row of SURVEY 8(f): the HIP encoder, called through the C ABI
(pypolar_amd.PolarEncoder -> pce_encode), against (a) golden vectors of the unmodified reference's
ButterflyFipPacked::encode_vector (tests/golden/enc.bin), (b) the oracle on seeded inputs incl. random frozen
sets, (c) size-independent properties at the BASELINE batch size.  Bar: bit-exact code words and checksums."""
import numpy as np
import pytest

import golden_io as gio
import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _encoder(N, frozen, crc_size=0, crc_type="crc", systematic=True):
    import pypolar_amd
    e = pypolar_amd.PolarEncoder(N, frozen)
    e.setErrorDetection(crc_size, crc_type)
    e.setSystematic(systematic)
    return e


def test_hip_encoder_matches_reference_vectors(built_library):
    rows = 0
    for fx in gio.load_enc():
        e = _encoder(fx.N, fx.frozen, fx.crc_size, fx.crc_type, fx.systematic)
        info = fx.info_in.copy()
        cw = e.encode_batch(info)
        assert np.array_equal(cw, fx.cw), (fx.name, fx.N, fx.K, fx.systematic)
        assert np.array_equal(info, fx.info_after), fx.name      # checksum written in place, like the reference
        rows += fx.count
    assert rows > 400


def test_encode_vector_surface(built_library):
    """pypolar.PolarEncoder.encode_vector drop-in (encoder_python.cc:41-60)"""
    fx = [f for f in gio.load_enc() if f.N == 1024 and f.crc_size == 24 and f.systematic][0]
    e = _encoder(fx.N, fx.frozen, 24, "crcnr")
    for r in range(4):
        a = fx.info_in[r].copy()
        cw = e.encode_vector(a)
        assert cw.dtype == np.uint8 and cw.shape == (128,)
        assert np.array_equal(cw, fx.cw[r])
        assert np.array_equal(a, fx.info_after[r])               # caller's array carries the CRC afterwards
        cw2 = e.encode_vector([int(x) for x in fx.info_in[r]])     # forcecast path: converted copy
        assert np.array_equal(cw2, fx.cw[r])


@pytest.mark.parametrize("N", [8, 16, 32, 64, 256, 1024, 2048, 4096, 16384, 32768])
def test_random_frozen_sets_vs_oracle(built_library, N):
    rng = np.random.default_rng(100 + N)
    for trial in range(4):
        K = int(rng.integers(1, N))
        if trial == 0:
            K = N - 1
        frozen = np.sort(rng.choice(N, N - K, replace=False)).astype(np.uint32)
        nb = (K + 7) // 8
        det = 0
        if trial % 2 == 1 and nb >= 8:
            det = [8, 11, 24, 32][int(rng.integers(0, 4))]
        typ = "crcnr" if det in (11, 24) else "crc"
        for sysm in (True, False):
            e = _encoder(N, frozen, det, typ, sysm)
            B = 37 if N <= 4096 else 5
            info = rng.integers(0, 256, (B, nb), dtype=np.uint8)
            if K % 8:
                info[:, -1] &= (0xFF << (8 - K % 8)) & 0xFF
            mine = info.copy()
            cw = e.encode_batch(mine)
            for r in range(B):
                i2, c2 = ob.encode(N, frozen, det, info[r], sysm)
                assert np.array_equal(cw[r], c2), (N, K, det, sysm, r)
                assert np.array_equal(mine[r], i2), (N, K, det, sysm, r)


def test_full_batch_properties_and_round_trip(built_library):
    """cfg3 at the BASELINE batch (2^20 rows, device resident): systematic property, linearity, a sample against
    the oracle, and encode -> BPSK -> decode with the product decoder gives the information words back."""
    import torch
    import pypolar_amd
    fx = gio.load_named("cfg3_n1024_k512_l8_crc24_awgn.bin")
    N, K, nb = fx.N, fx.K, (fx.K + 7) // 8
    B = 1 << 20
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    info = torch.randint(0, 256, (B, nb), dtype=torch.uint8, device=dev, generator=g)
    raw = info.clone()
    e = _encoder(N, fx.frozen, 24, "crcnr")
    cw = e.encode_batch(info)                      # fills the CRC fields of `info` in place
    torch.cuda.synchronize()
    assert torch.equal(info[:, : nb - 3], raw[:, : nb - 3])
    # sample vs the oracle
    idx = np.random.default_rng(3).integers(0, B, 64)
    h_raw, h_info, h_cw = raw[idx].cpu().numpy(), info[idx].cpu().numpy(), cw[idx].cpu().numpy()
    for r in range(idx.size):
        i2, c2 = ob.encode(N, fx.frozen, 24, h_raw[r], True)
        assert np.array_equal(h_info[r], i2) and np.array_equal(h_cw[r], c2)
    # systematic: the code bits on the information positions are the information bits
    shifts = torch.arange(7, -1, -1, device=dev, dtype=torch.uint8)
    cbits = ((cw.unsqueeze(-1) >> shifts) & 1).reshape(B, N)
    ibits = ((info.unsqueeze(-1) >> shifts) & 1).reshape(B, 8 * nb)[:, :K]
    mask = torch.ones(N, dtype=torch.bool)
    mask[torch.from_numpy(fx.frozen.astype(np.int64))] = False
    assert torch.equal(cbits[:, mask.to(dev)], ibits)
    # linearity over GF(2) (encoder and CRC are linear, zero maps to zero)
    half = B // 2
    x = (raw[:half] ^ raw[half:]).contiguous()
    cwx = e.encode_batch(x)
    assert torch.equal(cwx, cw[:half] ^ cw[half:])
    # round trip through the product decoder, noiseless BPSK
    d = pypolar_amd.PolarDecoder(N, 8, fx.frozen, "float")
    d.setErrorDetection(24, "crcnr")
    step = 1 << 18
    for s in range(0, B, step):
        llr = 1.0 - 2.0 * cbits[s:s + step].float()
        out, ok = d.decode_batch(llr)
        assert bool(ok.all())
        assert torch.equal(out, info[s:s + step])


def test_non_systematic_round_trip(built_library):
    import pypolar_amd
    fx = [f for f in gio.load_enc() if f.N == 1024 and f.crc_size == 11 and not f.systematic][0]
    e = _encoder(fx.N, fx.frozen, 11, "crcnr", systematic=False)
    rng = np.random.default_rng(9)
    info = rng.integers(0, 256, (256, (fx.K + 7) // 8), dtype=np.uint8)
    cw = e.encode_batch(info)
    bits = np.unpackbits(cw, axis=1)
    d = pypolar_amd.PolarDecoder(fx.N, 8, fx.frozen, "float")
    d.setErrorDetection(11, "crcnr")
    d.setSystematic(False)
    out, ok = d.decode_batch((1.0 - 2.0 * bits).astype(np.float32))
    assert ok.all() and np.array_equal(out, info)


def test_empty_and_single_row_batches(built_library):
    fx = [f for f in gio.load_enc() if f.N == 1024 and f.crc_size == 24 and f.systematic][0]
    e = _encoder(fx.N, fx.frozen, 24, "crcnr")
    out = e.encode_batch(np.zeros((0, 64), np.uint8))
    assert out.shape == (0, 128)
    one = fx.info_in[:1].copy()
    assert np.array_equal(e.encode_batch(one), fx.cw[:1]) and np.array_equal(one, fx.info_after[:1])
    nine = fx.info_in[:9].copy()          # one full wave pass of 8 code words + 1
    assert np.array_equal(e.encode_batch(nine), fx.cw[:9])
