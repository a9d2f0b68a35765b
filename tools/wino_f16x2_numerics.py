"""Numerical model (numpy, CPU) of a dilated k-tap conv layer computed in the WINOGRAD domain on the fp16 matrix cores with two-term
operands -- the arithmetic question behind the next step for the decoder trunk (docs/HISTORY.md 12): the exact-fp32 path already runs its
ResBlock layers as segmented F(2,3) / F(2,2) (resblock_wino_kernel: 0.67-0.73 of the direct form's matrix products), the default
two-term fp16 path (conv_bf3.hip MATH 1, docs/HISTORY.md 5f) runs the direct form.  Would the transform-domain operands survive the split?

  python tools/wino_f16x2_numerics.py

Model, as tools/split_numerics.py: an "MFMA" forms the 16 products of a K block exactly, sums them exactly and adds them to an fp32
accumulator with one rounding.  Taps are cut into segments of 3 / 2 as model.hip wino_split does (k = 3 -> 3; 7 -> 3 + 2 + 2;
11 -> 3 + 3 + 3 + 2).  Input transform in fp32 (one rounding per transformed value), weight transform in float64 rounded once to fp32,
then the shipped split: activation hi = fp16(v), lo' = fp16((v - hi) 2^11); weights scaled by one power of two per conv (max over all
transformed planes into [2^13, 2^14)), P0 / P1 / P2 = P0 2^-11; three products per fp32 product; the four (three) transform-domain
accumulators of all segments of a kind are shared; output transform in fp32."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import split_numerics as sn   # noqa: E402


def wino_split(k):
    n3, n2 = (k // 3, 0) if k % 3 == 0 else ((k - 2) // 3, 1) if k % 3 == 2 else ((k - 4) // 3, 2)
    return n3, n2


def f32(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float64)


def conv_truth(x, w, dil):
    """x [Cin][N + (k-1) dil] float64, w [Cout][k][Cin] -> [Cout][N]"""
    Cout, k, Cin = w.shape
    N = x.shape[1] - (k - 1) * dil
    y = np.zeros((Cout, N))
    for j in range(k):
        y += w[:, j, :] @ x[:, j * dil:j * dil + N]
    return y


def fp32_chain(x, w, dil):
    Cout, k, Cin = w.shape
    N = x.shape[1] - (k - 1) * dil
    acc = np.zeros((Cout, N), np.float32)
    for j in range(k):
        for c in range(Cin):
            acc = (acc.astype(np.float64) + w[:, j, c:c + 1] * x[c:c + 1, j * dil:j * dil + N]).astype(np.float32)
    return acc.astype(np.float64)


def direct_f16x2(x, w, dil):
    """the shipped form: K = (tap, channel), three products per fp32 product"""
    Cout, k, Cin = w.shape
    N = x.shape[1] - (k - 1) * dil
    A = w.reshape(Cout, k * Cin).astype(np.float32)
    B = np.concatenate([x[:, j * dil:j * dil + N] for j in range(k)], axis=0).astype(np.float32)
    Ws, down = sn.split_f16x2_weight(A)
    return sn.mfma_sum(Ws, sn.split_f16x2_act(B), sn.H2_PAIRS, k * Cin).astype(np.float64) * down


def wino_planes(x, w, dil):
    """-> list of (kind, U [nt][Cout][Cin] float64 (rounded to fp32), D [nt][Cin][N/2] (fp32 values)) per segment; outputs are pairs
    (n, n + dil) -- positions are processed in blocks of 2 dil: first dil positions = y0 of their pairs, next dil = y1."""
    Cout, k, Cin = w.shape
    N = x.shape[1] - (k - 1) * dil
    assert N % (2 * dil) == 0
    base = (np.arange(N // (2 * dil))[:, None] * 2 * dil + np.arange(dil)[None, :]).reshape(-1)     # y0 positions
    n3, n2 = wino_split(k)
    segs, j0 = [], 0
    for _ in range(n3):
        g = w[:, j0:j0 + 3, :]
        U = np.stack([g[:, 0], (g[:, 0] + g[:, 1] + g[:, 2]) / 2, (g[:, 0] - g[:, 1] + g[:, 2]) / 2, g[:, 2]])
        xs = [x[:, base + (j0 + t) * dil] for t in range(4)]
        D = np.stack([f32(xs[0] - xs[2]), f32(xs[1] + xs[2]), f32(xs[2] - xs[1]), f32(xs[1] - xs[3])])
        segs.append((3, f32(U), D)); j0 += 3
    for _ in range(n2):
        g = w[:, j0:j0 + 2, :]
        U = np.stack([g[:, 0], g[:, 0] + g[:, 1], g[:, 1]])
        xs = [x[:, base + (j0 + t) * dil] for t in range(3)]
        D = np.stack([f32(xs[0] - xs[1]), xs[1], f32(xs[1] - xs[2])])
        segs.append((2, f32(U), D)); j0 += 2
    return segs, base, N


def wino(x, w, dil, arithmetic):
    """arithmetic: 'f32' (exact products, fp32 accumulation per K block: the shipped exact-fp32 Winograd path's model) | 'f16x2'"""
    segs, base, N = wino_planes(x, w, dil)
    Cout = w.shape[0]
    if arithmetic == "f16x2":                      # one power of two for the whole conv, from the largest transformed weight
        mx = max(float(np.abs(U).max()) for _, U, _ in segs)
        up = 2.0 ** (14 - np.frexp(mx)[1])
    acc = {3: [np.zeros((Cout, base.size), np.float32) for _ in range(4)], 2: [np.zeros((Cout, base.size), np.float32) for _ in range(3)]}
    for kind, U, D in segs:
        for t in range(U.shape[0]):
            A, B = U[t].astype(np.float32), D[t].astype(np.float32)
            K = A.shape[1]
            if arithmetic == "f32":
                As, Bs, pairs = [A.astype(np.float64)], [B.astype(np.float64)], [(0, 0)]
            else:
                ws = (A * np.float32(up)).astype(np.float32)
                p0 = ws.astype(np.float16); p1 = (ws - p0.astype(np.float32)).astype(np.float32).astype(np.float16)
                p2 = (p0.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16)
                As, Bs, pairs = [p.astype(np.float64) for p in (p0, p1, p2)], sn.split_f16x2_act(B), sn.H2_PAIRS
            a = acc[kind][t]
            for k0 in range(0, K, 16):
                for i, j in pairs:
                    a = (a.astype(np.float64) + As[i][:, k0:k0 + 16] @ Bs[j][k0:k0 + 16, :]).astype(np.float32)
            acc[kind][t] = a
    down = np.float32(1.0 / up) if arithmetic == "f16x2" else np.float32(1.0)
    m3 = [a * down for a in acc[3]]; m2 = [a * down for a in acc[2]]            # (a power of two: exact)
    y0 = ((m3[0] + m3[1]) + m3[2]) + (m2[0] + m2[1])
    y1 = ((m3[1] - m3[2]) - m3[3]) + (m2[1] - m2[2])
    y = np.zeros((Cout, N))
    y[:, base] = y0.astype(np.float64); y[:, base + dil] = y1.astype(np.float64)
    return y


def study(k, dil, sx, seed=0, Cin=128, Cout=64, N=240):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((Cout, k, Cin)) * 0.03).astype(np.float32).astype(np.float64)
    x = (rng.standard_normal((Cin, N + (k - 1) * dil)) * rng.uniform(0.3, 3, (Cin, 1)) * sx).astype(np.float32).astype(np.float64)
    truth = conv_truth(x, w, dil)
    s = np.sqrt((truth ** 2).mean())
    err = lambda y: float(np.sqrt(((y - truth) ** 2).mean()) / s)   # noqa: E731
    return {"fp32 chain": err(fp32_chain(x, w, dil)), "direct f16x2 (shipped)": err(direct_f16x2(x, w, dil)),
            "winograd exact-fp32 (shipped f32 path)": err(wino(x, w, dil, "f32")), "winograd f16x2": err(wino(x, w, dil, "f16x2"))}


def main():
    print("relative rms error against float64; 128 -> 64 channels, weights ~ N(0, 0.03), activations ~ N(0, 0.3..3) x scale")
    for k, dil in ((3, 1), (3, 5), (7, 3), (11, 1)):
        for sx in (1.0, 0.01):
            r = study(k, dil, sx)
            n3, n2 = wino_split(k)
            print(f"k = {k:2d} dil = {dil} x scale {sx:5.2f} ({4 * n3 + 3 * n2} matrix products per output pair instead of {2 * k}): "
                  + "   ".join(f"{n}: {e:.2e}" for n, e in r.items()))


if __name__ == "__main__":
    main()
