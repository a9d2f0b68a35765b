"""Numerical model (numpy, CPU) of fp32 dot products computed on narrow-operand matrix cores with split operands, against
float64 -- the study behind conv_bf3.hip and behind the open question whether TWO fp16 terms (3 products,
half the matrix instructions of the shipped 3 x bf16 / 6 products scheme) could serve.

  python tools/split_numerics.py

Each "MFMA" is modelled as: the 16 products of a K block formed exactly, summed exactly, added to an fp32 accumulator with
one rounding -- optimistic about the hardware's internal adder, identical for all schemes, so the comparison stands.
Findings: (1) 3 x bf16 (truncation split: exact) with the 6 products of order <= 2^-16 is as accurate as a sequential fp32
multiply-add chain; 3 products are not (3e-5).  (2) 2 x fp16 (round-to-nearest split) with 3 products matches fp32 only
while BOTH terms stay in fp16's normal range: activations below ~0.1 and all second terms of typical weights (|w| ~ 0.03)
go subnormal (error x2..x14), and if the matrix core flushed subnormal inputs the error would be 2e-4.  It would need
per-conv power-of-two weight scales and a per-layer activation scale with an overflow escape -- not built; the measured
prize (MFMAs halved, results wrong by design) is 3.76 -> 3.13 ms per utterance and 26.8 -> 18.8 ms at batch 8."""
import numpy as np


def trunc_bf16(x):
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split_bf16x3(x):
    x = x.astype(np.float32)
    h = trunc_bf16(x); r = (x - h).astype(np.float32); m = trunc_bf16(r); l = (r - m).astype(np.float32)
    assert np.all(h.astype(np.float64) + m + l == x)            # exact
    return [v.astype(np.float64) for v in (h, m, l)]


def split_f16x2(x, ftz=False):
    x = x.astype(np.float32)
    h1 = x.astype(np.float16); h2 = (x - h1.astype(np.float32)).astype(np.float32).astype(np.float16)
    if ftz:
        tiny = np.float16(6.104e-05)
        h1 = np.where(np.abs(h1) < tiny, np.float16(0), h1); h2 = np.where(np.abs(h2) < tiny, np.float16(0), h2)
    return [h1.astype(np.float64), h2.astype(np.float64)]


def mfma_sum(As, Bs, pairs, K, blk=16):
    acc = np.zeros((As[0].shape[0], Bs[0].shape[1]), np.float32)
    for k0 in range(0, K, blk):
        for i, j in pairs:
            acc = (acc.astype(np.float64) + As[i][:, k0:k0 + blk] @ Bs[j][k0:k0 + blk, :]).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    M, N, K = 64, 256, 1408                                      # a 128-channel, 11-tap conv
    print("scheme                               x scale   rel. rms error   (fp32 multiply-add chain)")
    for sx in (10.0, 1.0, 0.1, 0.01, 0.001):
        A = (rng.standard_normal((M, K)) * 0.03).astype(np.float32)
        B = (rng.standard_normal((K, N)) * rng.uniform(0.3, 3, (K, 1)) * sx).astype(np.float32)
        truth = A.astype(np.float64) @ B.astype(np.float64)
        s = np.sqrt((truth ** 2).mean())
        acc = np.zeros((M, N), np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1, :].astype(np.float64)).astype(np.float32)
        e32 = np.sqrt(((acc - truth) ** 2).mean()) / s
        rows = [("3 x bf16, 6 products (shipped)", split_bf16x3(A), split_bf16x3(B), [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]),
                ("3 x bf16, 3 products", split_bf16x3(A), split_bf16x3(B), [(1, 0), (0, 1), (0, 0)]),
                ("2 x fp16, 3 products", split_f16x2(A), split_f16x2(B), [(1, 0), (0, 1), (0, 0)]),
                ("2 x fp16, 3 products, subnormals flushed", split_f16x2(A, True), split_f16x2(B, True), [(1, 0), (0, 1), (0, 0)])]
        for name, As, Bs, pairs in rows:
            e = np.sqrt(((mfma_sum(As, Bs, pairs, K) - truth) ** 2).mean()) / s
            print(f"{name:42s} {sx:7.3f}   {e:.2e}        ({e32:.2e})")


if __name__ == "__main__":
    main()
