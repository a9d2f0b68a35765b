"""Numerical model (numpy, CPU) of fp32 dot products computed on narrow-operand matrix cores with split operands, against
float64 -- the study behind conv_bf3.hip (docs/HISTORY.md 5d) and behind the open question whether TWO fp16 terms (3 products,
half the matrix instructions of the shipped 3 x bf16 / 6 products scheme) could serve.

  python tools/split_numerics.py

Each "MFMA" is modelled as: the 16 products of a K block formed exactly, summed exactly, added to an fp32 accumulator with
one rounding -- optimistic about the hardware's internal adder, identical for all schemes, so the comparison stands.
Findings: (1) 3 x bf16 (truncation split: exact) with the 6 products of order <= 2^-16 is as accurate as a sequential fp32
multiply-add chain; 3 products are not (3e-5).  (2) 2 x fp16 (round-to-nearest split) with 3 products matches fp32 only
while BOTH terms stay in fp16's normal range: activations below ~0.1 and all second terms of typical weights (|w| ~ 0.03)
go subnormal (error x2..x14), and if the matrix core flushed subnormal inputs the error would be 2e-4.  It would need
per-conv power-of-two weight scales and a per-layer activation scale with an overflow escape.  (3) The form that shipped in
round 3 (conv_bf3.hip MATH 1, "f16x2") needs no activation scale: the activation's second term is kept as lo' = fp16((x - hi) * 2^11),
as large as x itself, and meets a third weight plane P2 = P0 * 2^-11; weights are scaled per conv so that max |w| lands in
[2^13, 2^14).  It beats the fp32 chain at every activation scale (rows "shipped") provided the matrix core honours fp16
subnormals in hi (gfx950 does: tests/test_parity_gpu.py::test_f16x2_conv_against_float64 scales the inputs down to 2^-20)."""
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


def split_f16x2_act(x, ftz=False):
    """conv_bf3.hip split8h: hi = fp16(x), lo' = fp16((x - hi) * 2^11)"""
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float32).astype(np.float16)
    if ftz:
        tiny = np.float16(6.104e-05)
        hi = np.where(np.abs(hi) < tiny, np.float16(0), hi); lo = np.where(np.abs(lo) < tiny, np.float16(0), lo)
    return [hi.astype(np.float64), lo.astype(np.float64)]


def split_f16x2_weight(w, ftz=False):
    """conv_bf3.hip bf3_pack(math 1): ws = w * 2^s with max |ws| in [2^13, 2^14); P0 = fp16(ws), P1 = fp16(ws - P0), P2 = P0 * 2^-11.
    Returns ([P0, P1, P2], 2^-s)."""
    w = w.astype(np.float32)
    mx = float(np.abs(w).max())
    up = np.float32(2.0 ** (14 - np.frexp(mx)[1])) if mx > 0 else np.float32(1.0)
    ws = (w * up).astype(np.float32)
    p0 = ws.astype(np.float16)
    p1 = (ws - p0.astype(np.float32)).astype(np.float32).astype(np.float16)
    p2 = (p0.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
    if ftz:
        tiny = np.float16(6.104e-05)
        p0, p1, p2 = (np.where(np.abs(t) < tiny, np.float16(0), t) for t in (p0, p1, p2))
    return [p0.astype(np.float64), p1.astype(np.float64), p2.astype(np.float64)], float(1.0 / up)


H2_PAIRS = [(2, 1), (1, 0), (0, 0)]     # (weight plane, activation plane): P2 x lo', P1 x hi, P0 x hi -- smallest term first


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
        for ftz in (False, True):
            Ws, down = split_f16x2_weight(A, ftz)
            y = mfma_sum(Ws, split_f16x2_act(B, ftz), H2_PAIRS, K).astype(np.float64) * down
            e = np.sqrt(((y.astype(np.float32) - truth) ** 2).mean()) / s
            print(f"{'2 x fp16 scaled, 3 products (shipped)' + (', subnormals flushed' if ftz else ''):42s} {sx:7.3f}   {e:.2e}        ({e32:.2e})")


if __name__ == "__main__":
    main()
