"""Frozen-set construction (host side, one-time): the Bhattacharyya-bound construction of the reference.

Mirrors ``pypolar.frozen_bits(blockLength, infoLength, designSNR, "BB")``
(reference: src/polarcode/construction/constructor.cpp:41-63, bhattacharrya.cpp:39-81): the channel
parameter recursion in double precision from a float32 initial value, a STABLE descending sort
(trackingSorter::stableSortDescending, arrayfuncs.cpp:90-106), the first N-K indices frozen.
Checked against the frozen sets stored in the golden fixtures (tests/test_host.py).
"""
import math
import os

import numpy as np


def bhattacharyya_parameters(N, K, design_snr_db):
    lin = np.float32(math.pow(10.0, float(np.float32(design_snr_db)) / 10.0))
    z0 = float(np.float32(math.exp(-2.0 * float(lin) * K / N)))   # stored in a float member
    z = np.zeros(N, np.float64)
    z[0] = z0
    n = int(round(math.log2(N)))
    for stage in range(n - 1, -1, -1):
        B = 1 << stage
        j = np.arange(0, N, 2 * B)
        T = z[j].copy()
        z[j + B] = T * T
        z[j] = 2 * T - z[j + B]
    return z


def frozen_bits(blockLength, infoLength, designSNR=0.0, constructor_type="BB"):
    N, K = int(blockLength), int(infoLength)
    t = str(constructor_type).lower()
    if "be" in t:
        return beta_expansion_frozen_bits(N, K)
    if "5g" in t:
        return nr_sequence_frozen_bits(N, K)
    z = bhattacharyya_parameters(N, K, designSNR)
    order = np.argsort(-z, kind="stable")     # stable descending: ties keep ascending index order
    return np.sort(order[: N - K]).astype(np.uint32)


def beta_expansion_frozen_bits(N, K):
    """``frozen_bits(N, K, _, "BE")``: polarisation-weight construction with beta = 2^(1/4)
    (reference: src/polarcode/construction/betaexpansion.cpp:44-78): weight(i) = sum_j bit_j(i) * beta^j, the
    N-K smallest weights are frozen."""
    n = int(round(math.log2(N)))
    beta = math.pow(2.0, 1.0 / 4.0)
    w = [math.pow(beta, float(j)) for j in range(n)]
    weights = np.zeros(N, np.float64)
    for i in range(N):
        acc = 0.0
        for j in range(n):
            acc += w[j] * ((i >> j) & 1)
        weights[i] = acc
    order = np.argsort(weights, kind="stable")
    return np.sort(order[: N - K]).astype(np.uint32)


_NR_SEQUENCE = None


def nr_polar_sequence():
    """The polar sequence of 3GPP TS 38.212 Table 5.3.1.2-1 (positions in ascending reliability), as the reference's
    "5G" construction uses it.  Shipped as data (``data/nr_polar_sequence_1024.u16``, produced by running the
    reference: ``oracle/_ref/ref_harness nrseq``)."""
    global _NR_SEQUENCE
    if _NR_SEQUENCE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "nr_polar_sequence_1024.u16")
        q = np.fromfile(path, dtype="<u2").astype(np.uint32)
        assert q.size == 1024 and np.array_equal(np.sort(q), np.arange(1024))
        _NR_SEQUENCE = q
    return _NR_SEQUENCE


def nr_sequence_frozen_bits(N, K):
    """``frozen_bits(N, K, _, "5G")`` (reference: src/polarcode/construction/fiveGList.cpp:20-43): the first N - K
    entries of the 1024-entry sequence, sorted.  Like the reference, entries are NOT filtered by ``< N``, so the
    result is only a valid frozen set for N = 1024; N > 1024 raises as the reference does."""
    if N <= 0 or (1 << int(math.log2(N))) != N:
        raise ValueError("new blockLength is not a power of 2!")
    if N > 1024:
        raise ValueError("5G standard does not allow for block size N > 1024!")
    if N < K:
        raise ValueError(f"Invalid polar code({N}, {K})")
    return np.sort(nr_polar_sequence()[: N - K]).astype(np.uint32)
