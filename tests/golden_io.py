"""Loader for the golden fixtures under tests/golden/ (format: oracle/ref_harness.cpp header).

A file may hold several concatenated fixtures (the sweep files do); ``load`` yields one
``Fixture`` per record.  Fixtures are DATA produced by the unmodified reference library
(oracle/gen_golden.sh); nothing here reads /root/reference.
"""
import glob
import os
from dataclasses import dataclass

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAGIC = 0x31584650


@dataclass
class Fixture:
    name: str
    N: int
    K: int
    L: int
    frozen: np.ndarray      # uint32[nF], ascending
    crc_size: int
    crc_nr: bool
    dectype: str            # "float" | "mixed" | "char"
    llr: np.ndarray         # float32[count, N]
    info: np.ndarray        # uint8[count, nb]   payload incl. checksum, as encoded
    out: np.ndarray         # uint8[count, nb]   the reference decoder's output
    ok: np.ndarray          # uint8[count]
    npaths: np.ndarray      # uint8[count]
    metrics: np.ndarray     # float32[count, L]  reference path metrics, best first
    systematic: bool = True  # False: setSystematic(false) on encoder and decoder
    i8_input: bool = False   # True: the (int8-valued) row went through decode_vector(const char*)

    @property
    def count(self):
        return self.llr.shape[0]

    @property
    def crc_type(self):
        return "crcnr" if self.crc_nr else "crc"


def load(path):
    raw = np.fromfile(path, dtype=np.uint8)
    pos, idx, out = 0, 0, []
    base = os.path.basename(path)
    while pos < raw.size:
        hdr = raw[pos:pos + 40].view(np.uint32)
        assert hdr[0] == MAGIC, f"bad magic in {path} @ {pos}"
        _, N, K, L, nF, count, nb, crcsize, crcnr, dectype = (int(x) for x in hdr)
        pos += 40
        frozen = raw[pos:pos + 4 * nF].view(np.uint32).copy()
        pos += 4 * nF
        rec = 4 * N + nb + nb + 1 + 1 + 4 * L
        body = raw[pos:pos + rec * count].reshape(count, rec)
        pos += rec * count
        o = 0
        llr = body[:, o:o + 4 * N].copy().view(np.float32).reshape(count, N); o += 4 * N
        info = body[:, o:o + nb].copy(); o += nb
        outb = body[:, o:o + nb].copy(); o += nb
        ok = body[:, o].copy(); o += 1
        npaths = body[:, o].copy(); o += 1
        metrics = body[:, o:o + 4 * L].copy().view(np.float32).reshape(count, L)
        kind = {0: "float", 2: "mixed", 3: "char"}[dectype & 0xFF]
        out.append(Fixture(f"{base}#{idx}", N, K, L, frozen, crcsize, bool(crcnr), kind, llr, info, outb, ok, npaths,
                           metrics, systematic=not (dectype & 0x100), i8_input=bool(dectype & 0x200)))
        idx += 1
    return out


def all_files(pattern="*.bin"):
    """decoder fixtures ('PFX1' records); the encoder fixture enc.bin has its own loader"""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, pattern)) if os.path.basename(f) not in ("enc.bin", "punc.bin"))


ENC_MAGIC = 0x31454650


@dataclass
class EncFixture:
    name: str
    N: int
    K: int
    systematic: bool
    frozen: np.ndarray      # uint32[nF]
    crc_size: int
    crc_nr: bool
    info_in: np.ndarray     # uint8[count, nb]  what the caller handed to encode_vector
    info_after: np.ndarray  # uint8[count, nb]  the same buffer afterwards (checksum written in place)
    cw: np.ndarray          # uint8[count, N/8] packed code word, bit i = byte i/8, bit 7 - i%8

    @property
    def count(self):
        return self.cw.shape[0]

    @property
    def crc_type(self):
        return "crcnr" if self.crc_nr else "crc"


def load_enc(path=None):
    path = path or os.path.join(GOLDEN_DIR, "enc.bin")
    raw = np.fromfile(path, dtype=np.uint8)
    pos, idx, out = 0, 0, []
    while pos < raw.size:
        hdr = raw[pos:pos + 40].view(np.uint32)
        assert hdr[0] == ENC_MAGIC, f"bad magic in {path} @ {pos}"
        _, N, K, sysm, nF, count, nb, crcsize, crcnr, _ = (int(x) for x in hdr)
        pos += 40
        frozen = raw[pos:pos + 4 * nF].view(np.uint32).copy()
        pos += 4 * nF
        rec = 2 * nb + N // 8
        body = raw[pos:pos + rec * count].reshape(count, rec)
        pos += rec * count
        out.append(EncFixture(f"enc.bin#{idx}", N, K, bool(sysm), frozen, crcsize, bool(crcnr),
                              body[:, :nb].copy(), body[:, nb:2 * nb].copy(), body[:, 2 * nb:].copy()))
        idx += 1
    return out


def load_named(name):
    fx = load(os.path.join(GOLDEN_DIR, name))
    assert len(fx) == 1
    return fx[0]


PUNC_MAGIC = 0x31504650


@dataclass
class PuncFixture:
    name: str
    M: int                   # punctured block length
    N: int                   # parent block length
    frozen: np.ndarray       # uint32[nF]
    positions: np.ndarray    # uint32[M]  Puncturer::blockOutputPositions()
    x: np.ndarray            # float32[count, N]
    punctured: np.ndarray    # float32[count, M]   puncture(x)
    depunctured: np.ndarray  # float32[count, N]   depuncture(punctured)
    packed_in: np.ndarray    # uint8[count, N/8]
    packed_out: np.ndarray   # uint8[count, M/8]   puncturePacked(packed_in)  (only meaningful if M % 8 == 0)

    @property
    def count(self):
        return self.x.shape[0]


def load_punc(path=None):
    path = path or os.path.join(GOLDEN_DIR, "punc.bin")
    raw = np.fromfile(path, dtype=np.uint8)
    pos, idx, out = 0, 0, []
    while pos < raw.size:
        hdr = raw[pos:pos + 40].view(np.uint32)
        assert hdr[0] == PUNC_MAGIC, f"bad magic in {path} @ {pos}"
        _, M, N, K, nF, count = (int(x) for x in hdr[:6])
        pos += 40
        frozen = raw[pos:pos + 4 * nF].view(np.uint32).copy(); pos += 4 * nF
        positions = raw[pos:pos + 4 * M].view(np.uint32).copy(); pos += 4 * M
        rec = 4 * N + 4 * M + 4 * N + N // 8 + M // 8
        body = raw[pos:pos + rec * count].reshape(count, rec); pos += rec * count
        o = 0
        x = body[:, o:o + 4 * N].copy().view(np.float32).reshape(count, N); o += 4 * N
        pu = body[:, o:o + 4 * M].copy().view(np.float32).reshape(count, M); o += 4 * M
        de = body[:, o:o + 4 * N].copy().view(np.float32).reshape(count, N); o += 4 * N
        pi = body[:, o:o + N // 8].copy(); o += N // 8
        po = body[:, o:o + M // 8].copy()
        out.append(PuncFixture(f"punc.bin#{idx}", M, N, frozen, positions, x, pu, de, pi, po))
        idx += 1
    return out
