"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboracle.so")
_lib = None

DET = {(0, False): 0, (0, True): 0, (8, False): 8, (8, True): 8, (11, True): 11, (11, False): 11,
       (24, True): 24, (24, False): 24, (32, False): 32, (32, True): 32}


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_ROOT, "oracle", "polar_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle"), "liboracle.so"])
        _lib = C.CDLL(_SO)
        _lib.po_decode_batch.restype = C.c_int
        _lib.po_decode_batch.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
        _lib.po_det_check.restype = C.c_int
        _lib.po_det_check.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _lib.po_det_calc.restype = C.c_uint32
        _lib.po_det_calc.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _lib.po_det_generate.argtypes = [C.c_int, C.c_void_p, C.c_int]
        _lib.po_encode.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]
        _lib.po_tree_stats.restype = C.c_int
        _lib.po_tree_stats.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_int, C.c_void_p]
        _lib.po_f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        _lib.po_g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        _lib.po_sc_decode_bits.restype = C.c_int
        _lib.po_sc_decode_bits.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        _lib.po_scl_decode_paths.restype = C.c_int
        _lib.po_scl_decode_paths.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        _lib.po_scl_trace.restype = C.c_int
        _lib.po_scl_trace.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def decode_batch(N, L, frozen, det, llr, systematic=True, dectype="float"):
    """-> (out uint8[B, nb], ok uint8[B], metrics float32[B, L])"""
    frozen = np.ascontiguousarray(frozen, dtype=np.uint32)
    llr = np.ascontiguousarray(llr, dtype=np.float32).reshape(-1, N)
    B = llr.shape[0]
    nb = (N - frozen.size + 7) // 8
    out = np.zeros((B, nb), np.uint8)
    ok = np.zeros(B, np.uint8)
    met = np.zeros((B, max(L, 1)), np.float32)
    rc = lib().po_decode_batch(N, L, _p(frozen), frozen.size, det, int(systematic),
                               {"mixed": 2, "char": 3}.get(dectype, 0), _p(llr), B, _p(out), _p(ok), _p(met))
    if rc != 0:
        raise ValueError("oracle: the reference would throw for this frozen set (SURVEY Q6)")
    return out, ok, met


def det_check(det, data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return bool(lib().po_det_check(det, _p(data), data.size))


def det_calc(det, data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().po_det_calc(det, _p(data), data.size))


def det_generate(det, data):
    data = np.ascontiguousarray(data, dtype=np.uint8).copy()
    lib().po_det_generate(det, _p(data), data.size)
    return data


def encode(N, frozen, det, info, systematic=True):
    """info uint8[nb] (checksum tail is overwritten) -> (info_with_crc, codeword uint8[N/8])"""
    frozen = np.ascontiguousarray(frozen, dtype=np.uint32)
    info = np.ascontiguousarray(info, dtype=np.uint8).copy()
    cw = np.zeros(N // 8, np.uint8)
    lib().po_encode(N, _p(frozen), frozen.size, det, int(systematic), _p(info), _p(cw))
    return info, cw


def tree_stats(N, frozen, list_decoder):
    frozen = np.ascontiguousarray(frozen, dtype=np.uint32)
    cnt = np.zeros(2, np.uint32)
    rc = lib().po_tree_stats(N, _p(frozen), frozen.size, int(list_decoder), _p(cnt))
    return rc, int(cnt[0]), int(cnt[1])


def scl_trace(N, L, frozen, llr_row, cap=4096):
    frozen = np.ascontiguousarray(frozen, dtype=np.uint32)
    a = np.ascontiguousarray(llr_row, dtype=np.float32)
    tr = np.zeros((cap, 1 + L), np.float32)
    n = lib().po_scl_trace(N, L, _p(frozen), frozen.size, _p(a), _p(tr), cap)
    return tr[:n]


def punc_positions(M, frozen):
    """-> (positions uint32[M], parent block length) or raises IndexError like the reference (std::out_of_range)"""
    frozen = np.ascontiguousarray(frozen, dtype=np.uint32)
    N = 1
    while N < M:
        N <<= 1
    pos = np.zeros(M, np.uint32)
    par = C.c_uint()
    L = lib()
    L.po_punc_positions.restype = C.c_int
    rc = L.po_punc_positions(C.c_uint(M), _p(frozen), C.c_uint(frozen.size), _p(pos), C.byref(par))
    if rc:
        raise IndexError("Number of required puncturing positions exceeds frozen bit positions!")
    return pos, int(par.value)


def puncture_f32(pos, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(pos.size, np.float32)
    lib().po_puncture_f32(_p(pos), C.c_uint(pos.size), _p(x), _p(out))
    return out


def depuncture_f32(pos, N, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(N, np.float32)
    lib().po_depuncture_f32(_p(pos), C.c_uint(pos.size), C.c_uint(N), _p(x), _p(out))
    return out


def puncture_packed(pos, x):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.zeros(pos.size // 8, np.uint8)
    lib().po_puncture_packed(_p(pos), C.c_uint(pos.size), _p(x), _p(out))
    return out
