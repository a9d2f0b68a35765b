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


def test_two_scaled_fp16_terms_hold_an_fp32_value_to_23_bits():
    """conv_bf3.hip MATH 1: x ~= hi + 2^-11 lo' with hi = fp16(x), lo' = fp16((x - hi) 2^11) -- relative error <= 2^-22 (2^-23 but for
    the ties) wherever |x| is inside fp16's normal range, an absolute error <= 2^-36 below it; lo' never exceeds |x|."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(40000) * np.exp(rng.uniform(-9, 10, 40000))).astype(np.float32)
    x = x[np.abs(x) < 60000]
    hi, lo = sn.split_f16x2_act(x)
    rec = hi + lo * 2.0 ** -11
    err = np.abs(rec - x.astype(np.float64))
    normal = np.abs(x) >= 2.0 ** -14
    assert np.all(err[normal] <= np.abs(x[normal]) * 2.0 ** -22)
    assert np.all(err[~normal] <= 2.0 ** -36)
    assert np.all(np.abs(lo) <= np.abs(x.astype(np.float64)) * 1.0001 + 2.0 ** -13)


def test_scaled_weight_planes_of_the_two_term_form():
    """bf3_pack(math 1): the per-conv power of two brings max |w| into [2^13, 2^14); P0 + P1 holds ws to 2^-22 relative wherever
    |ws| >= 2^-3 and to 2^-25 absolute below; P2 is exactly P0 2^-11 above fp16's subnormals."""
    rng = np.random.default_rng(6)
    w = (rng.standard_normal((64, 256)) * 0.03 * np.exp(rng.uniform(-6, 0, (64, 1)))).astype(np.float32)
    (p0, p1, p2), down = sn.split_f16x2_weight(w)
    ws = w.astype(np.float64) / down
    assert 2.0 ** 13 <= np.abs(ws).max() < 2.0 ** 14
    err = np.abs(p0 + p1 - ws)
    big = np.abs(ws) >= 2.0 ** -3
    assert np.all(err[big] <= np.abs(ws[big]) * 2.0 ** -22) and np.all(err[~big] <= 2.0 ** -25)
    ok = np.abs(p0) >= 2.0 ** -3
    assert np.array_equal(p2[ok], p0[ok] * 2.0 ** -11)


def test_three_products_of_the_scaled_two_term_form_match_an_fp32_chain_at_every_activation_scale():
    rng = np.random.default_rng(7)
    M, N, K = 32, 64, 704
    A = (rng.standard_normal((M, K)) * 0.03).astype(np.float32)
    Ws, down = sn.split_f16x2_weight(A)
    for sx in (100.0, 1.0, 0.01, 0.001):
        B = (rng.standard_normal((K, N)) * rng.uniform(0.01, 3, (K, 1)) * sx).astype(np.float32)
        truth = A.astype(np.float64) @ B.astype(np.float64)
        s = np.sqrt((truth ** 2).mean())
        acc = np.zeros((M, N), np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64)).astype(np.float32)
        e32 = np.sqrt(((acc - truth) ** 2).mean()) / s
        y = (sn.mfma_sum(Ws, sn.split_f16x2_act(B), sn.H2_PAIRS, K).astype(np.float64) * down).astype(np.float32)
        e = np.sqrt(((y - truth) ** 2).mean()) / s
        assert e <= 1.0 * e32, (sx, e, e32)
        # the unscaled two-term split of round 2's study loses it once the second terms go subnormal
        if sx <= 0.001:
            naive = sn.mfma_sum(sn.split_f16x2(A), sn.split_f16x2(B), [(1, 0), (0, 1), (0, 0)], K)
            assert np.sqrt(((naive - truth) ** 2).mean()) / s > 5 * e


def test_winograd_domain_operands_survive_the_two_term_split():
    """tools/wino_f16x2_numerics.py (docs/HISTORY.md 12): the segmented F(2,3) / F(2,2) form of a dilated ResBlock conv -- input transform in
    fp32, weight transform in float64, then the shipped two-term fp16 split with ONE power-of-two scale per conv -- is at least as
    accurate as the shipped direct two-term form and as an fp32 multiply-add chain, with 4 n3 + 3 n2 instead of 2 k matrix products per
    output pair.  (A statement about the arithmetic only: no such kernel is shipped.)"""
    import wino_f16x2_numerics as wn
    assert [wn.wino_split(k) for k in (3, 7, 11, 5, 2)] == [(1, 0), (1, 2), (3, 1), (1, 1), (0, 1)]
    for k, dil, sx in ((3, 5, 1.0), (7, 3, 0.01), (11, 1, 1.0)):
        r = wn.study(k, dil, sx, seed=11, Cin=64, Cout=32, N=120)
        assert r["winograd f16x2"] <= 1.05 * r["direct f16x2 (shipped)"], (k, dil, sx, r)
        assert r["winograd f16x2"] <= r["fp32 chain"], (k, dil, sx, r)
        assert r["winograd exact-fp32 (shipped f32 path)"] <= r["winograd f16x2"] * 1.05, (k, dil, sx, r)
