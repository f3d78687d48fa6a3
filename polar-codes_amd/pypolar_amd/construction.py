"""Frozen-set construction (host side, one-time): the Bhattacharyya-bound construction of the reference.

Mirrors ``pypolar.frozen_bits(blockLength, infoLength, designSNR, "BB")``
(reference: src/polarcode/construction/constructor.cpp:41-63, bhattacharrya.cpp:39-81): the channel
parameter recursion in double precision from a float32 initial value, a STABLE descending sort
(trackingSorter::stableSortDescending, arrayfuncs.cpp:90-106), the first N-K indices frozen.
Checked against the frozen sets stored in the golden fixtures (tests/test_host.py).
"""
import math

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
        raise NotImplementedError("the 5G reliability table is not shipped; pass that frozen set explicitly "
                                  "(it is a plain index list)")
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
