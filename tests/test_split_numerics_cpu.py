"""CPU checks of the arithmetic model behind conv_bf3.hip (tools/split_numerics.py): the three-term bf16 split of an fp32
number is EXACT, and six of the nine cross products reproduce an fp32 dot product to the accuracy of a sequential fp32
multiply-add chain -- the claim the GPU parity tests then verify on the hardware."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import split_numerics as sn   # noqa: E402


def test_three_bf16_terms_represent_every_fp32_value_exactly():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000) * np.exp(rng.uniform(-20, 20, 20000)),
                        [0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 2.0 ** -100, 1 + 2.0 ** -23, 1 - 2.0 ** -24]]).astype(np.float32)
    # (exact down to |x| ~ 2^-110: below that the third term would be an fp32 subnormal, which bf16's 7-bit mantissa cannot hold)
    h, m, l = sn.split_bf16x3(x)                      # asserts hi + mid + lo == x internally
    for t in (h, m, l):                               # every term is a bf16 number: its low 16 mantissa bits are zero
        assert np.all((t.astype(np.float32).view(np.uint32) & 0xFFFF) == 0)
    # the terms shrink by >= 2^-8 each (that is what bounds the dropped products)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -7) and np.all(np.abs(l[nz]) <= np.abs(h[nz]) * 2.0 ** -15)


def test_six_products_match_an_fp32_multiply_add_chain():
    rng = np.random.default_rng(4)
    M, N, K = 32, 64, 704
    A = (rng.standard_normal((M, K)) * 0.03).astype(np.float32)
    B = (rng.standard_normal((K, N)) * rng.uniform(0.01, 3, (K, 1))).astype(np.float32)
    truth = A.astype(np.float64) @ B.astype(np.float64)
    s = np.sqrt((truth ** 2).mean())
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64)).astype(np.float32)
    e32 = np.sqrt(((acc - truth) ** 2).mean()) / s
    six = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]
    e6 = np.sqrt(((sn.mfma_sum(sn.split_bf16x3(A), sn.split_bf16x3(B), six, K) - truth) ** 2).mean()) / s
    e3 = np.sqrt(((sn.mfma_sum(sn.split_bf16x3(A), sn.split_bf16x3(B), six[3:], K) - truth) ** 2).mean()) / s
    assert e6 <= 1.2 * e32            # as accurate as the fp32 chain
    assert e3 > 10 * e32              # the three 2^-16 products are needed
