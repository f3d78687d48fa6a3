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
