"""Host-side error-detector arithmetic (pure numpy, one-time / small inputs): the checksum layouts of the
reference's detectors, used to BUILD test traffic (bench.py, examples).  The decode path itself verifies
checksums on the GPU from syndrome tables (csrc/program.cpp).

Reference: src/polarcode/errordetection/{crc8,crc11nr,crc24nrc,crc32}.cpp (generate())."""
import numpy as np

_PARAMS = {8: (8, 0x07, 1), 11: (11, 0x621, 2), 24: (24, 0xB2B117, 3)}


def _crc_msb(data, width, poly):
    top, mask, r = 1 << (width - 1), (1 << width) - 1, 0
    for byte in bytes(data):
        for b in range(7, -1, -1):
            fb = (1 if (r & top) else 0) ^ ((byte >> b) & 1)
            r = (r << 1) & mask
            if fb:
                r ^= poly
    return r


def _crc32c_le(data):
    r = 0
    for byte in bytes(data):
        r ^= byte
        for _ in range(8):
            r = (r >> 1) ^ (0x82F63B78 if (r & 1) else 0)
    return r


def field_bytes(size):
    return {0: 0, 8: 1, 11: 2, 24: 3, 32: 4}[size]


def generate(size, data):
    """Detector::generate(data, bytes): returns a copy of `data` with the checksum written into its tail."""
    d = np.array(data, dtype=np.uint8).copy()
    n = d.size
    if size == 0:
        return d
    if size in _PARAMS:
        w, poly, fb = _PARAMS[size]
        c = _crc_msb(d[: n - fb], w, poly)
        for k in range(fb):
            d[n - 1 - k] = (c >> (8 * k)) & 0xFF
        return d
    if size == 32:
        wds = n // 4 - 1
        c = _crc32c_le(d[: 4 * wds])
        d[4 * wds: 4 * wds + 4] = np.frombuffer(int(c).to_bytes(4, "little"), np.uint8)
        return d
    raise ValueError("unsupported detector size")


def check(size, data):
    d = np.array(data, dtype=np.uint8)
    return bool(np.array_equal(generate(size, d), d)) if size else True


def checksum_bit_matrix(size, nb):
    """GF(2) matrix M [8*(nb-fb), 8*fb]: checksum-field bits (MSB-first per byte) = message bits @ M mod 2.
    All supported detectors are linear with a zero initial value."""
    fb = field_bytes(size)
    nmsg = 8 * (nb - fb)
    M = np.zeros((nmsg, 8 * fb), np.uint8)
    for t in range(nmsg):
        msg = np.zeros(nb, np.uint8)
        msg[t >> 3] = 0x80 >> (t & 7)
        M[t] = np.unpackbits(generate(size, msg)[nb - fb:])
    return M


def _crc_msb_bits(data, nbits, width, poly):
    """CRC++ CalculateBits (CRC.h:613-645, :1059-1110): MSB-first over the first nbits bits of the byte stream"""
    top, mask, r = 1 << (width - 1), (1 << width) - 1, 0
    d = bytes(np.asarray(data, dtype=np.uint8))
    for t in range(nbits):
        bit = (d[t >> 3] >> (7 - (t & 7))) & 1
        fb = (1 if (r & top) else 0) ^ bit
        r = (r << 1) & mask
        if fb:
            r ^= poly
    return r


class Detector:
    """``pypolar.Detector(size, type)`` (reference: python/bindings/detector_python.cc:19-77) for the detectors of
    the accelerated path: dummy (size 0), CRC-8, CRC-11 (NR), CRC-24C (NR), CRC-32C.  Host-side utility."""

    def __init__(self, size, type):
        t = str(type).lower()
        if "crc" in t:
            if size not in (0, 6, 8, 11, 16, 24, 32):
                raise RuntimeError("CRC INVALID SIZE!")
            if size in (6, 16):
                raise NotImplementedError("CRC-6 / CRC-16 are not on the accelerated path")
        elif "cmac" in t:
            raise NotImplementedError("CMAC is not on the accelerated path")
        else:
            raise RuntimeError("Unknown Error detector requested!")
        self._size = int(size)

    def getCheckBitCount(self):
        return self._size

    def getType(self):
        return "DUMMY" if self._size == 0 else ("CRCNR" if self._size in (11, 24) else "CRC")

    @staticmethod
    def _arr(data):
        a = np.asarray(data)
        if a.ndim != 1:
            raise RuntimeError("Only ONE-dimensional vectors allowed!")
        return np.ascontiguousarray(a.astype(np.uint8))

    def calculate(self, data, bitsize):
        a = self._arr(data)
        if a.size * 8 < bitsize:
            raise RuntimeError("Provided number of bits is smaller than stated significant bits!")
        if self._size in _PARAMS:
            w, poly, _ = _PARAMS[self._size]
            if self._size == 8:                       # CRC8::calculate works on whole bytes (crc8.cpp:31-34)
                return _crc_msb(a[: bitsize // 8], w, poly)
            return _crc_msb_bits(a, int(bitsize), w, poly)
        if self._size == 32:                          # CRC32::calculate: gen(words, bits / 8) (crc32.cpp:28-31)
            return _crc32c_le(a[: 4 * (int(bitsize) // 8)])
        return 0

    def generate(self, data):
        a = self._arr(data)
        fb = (self._size + 8) // 8 if self._size % 8 else self._size // 8
        return generate(self._size, np.concatenate([a, np.zeros(fb, np.uint8)]))

    def check(self, data):
        return check(self._size, self._arr(data))
