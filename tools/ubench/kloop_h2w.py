#!/usr/bin/env python3
"""The Winograd-domain lab kernel (conv_h2w.hip) against float64 and against the direct pre-split kernel (conv_h2p.hip) on the trunk's shapes.
  python tools/ubench/kloop_h2w.py check | time"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from summertts_amd import engine as eng   # noqa: E402

F = int(os.environ.get("KLOOP_FRAMES", "668"))


def ref64(x, w, b, dil, res, slope_in):
    xa = np.where(x < 0, x * np.float32(slope_in), x).astype(np.float64)
    co, k, ci = w.shape
    L = x.shape[1]
    pad = dil * (k - 1) // 2
    xp = np.pad(xa, ((0, 0), (pad, pad)))
    y = sum(w[:, t, :].astype(np.float64) @ xp[:, t * dil:t * dil + L] for t in range(k)) + b[:, None]
    return y + res if res is not None else y


def check():
    rng = np.random.default_rng(0)
    bad = 0
    for (C_, k, dil, L) in ((128, 3, 1, 777), (128, 3, 3, 500), (128, 3, 5, 333), (128, 7, 1, 300), (128, 7, 3, 1500), (128, 11, 5, 300), (128, 11, 1, 190),
                            (256, 3, 1, 1029), (256, 11, 5, 97), (256, 7, 5, 131), (128, 3, 1, 5), (128, 5, 2, 211)):
        x = rng.standard_normal((C_, L)).astype(np.float32) * 1.5
        w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32)
        b = rng.standard_normal(C_).astype(np.float32)
        res = rng.standard_normal((C_, L)).astype(np.float32)
        r64 = ref64(x, w, b, dil, res, 0.1)
        yd, *_ = eng.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=0)
        y, y16, _ = eng.debug_conv_h2w(x, w, b, dil, res, 0.1, 0.1, members=2)
        e_dir = float(np.abs(yd - r64).max()); e_w = float(np.abs(y - r64).max())
        rms_dir = float(np.sqrt(((yd - r64) ** 2).mean())); rms_w = float(np.sqrt(((y - r64) ** 2).mean()))
        lre = np.where(y < 0, y * np.float32(0.1), y)
        e16 = float(np.abs(y16 - lre).max())
        ok = e_w <= 3.0 * e_dir + 1e-6 and rms_w <= 2.0 * rms_dir + 1e-8 and e16 <= 1e-6
        bad += not ok
        print(f"C={C_} k={k} d={dil} L={L}: max-abs error vs float64: direct {e_dir:.2e} winograd {e_w:.2e}; rms {rms_dir:.2e} / {rms_w:.2e}; x16 vs lrelu(y) {e16:.1e}  {'ok' if ok else 'FAIL'}", flush=True)
    print("CHECK", "FAILED" if bad else "passed")
    return bad


def time_all():
    rng = np.random.default_rng(0)
    for name, C_, k, dil, L in [("s2_k3", 128, 3, 1, 64 * F), ("s2_k3d3", 128, 3, 3, 64 * F), ("s2_k7d3", 128, 7, 3, 64 * F), ("s2_k11d5", 128, 11, 5, 64 * F), ("s2_k11d1", 128, 11, 1, 64 * F),
                                ("s1_k3", 256, 3, 1, 8 * F), ("s1_k11d5", 256, 11, 5, 8 * F), ("b_s2_k3", 128, 3, 1, 640 * F), ("b_s2_k11d5", 128, 11, 5, 640 * F)]:
        x = rng.standard_normal((C_, L)).astype(np.float32)
        w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32)
        b = rng.standard_normal(C_).astype(np.float32)
        res = rng.standard_normal((C_, L)).astype(np.float32)
        flops = 2.0 * C_ * C_ * k * L
        for members in (1, 3):
            line = f"{name:10s} members={members}:"
            for tile in (0, 3):
                *_o, ms = eng.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=tile, members=members, iters=-10)
                line += f"  direct t{tile} {ms * 1e3:7.1f} us {members * flops / ms / 1e9:6.1f} TF"
            *_o, ms = eng.debug_conv_h2w(x, w, b, dil, res, 0.1, 0.1, members=members, iters=-10)
            line += f"  | winograd {ms * 1e3:7.1f} us {members * flops / ms / 1e9:6.1f} TF"
            print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    sys.exit((1 if check() else 0) if what == "check" else (time_all() or 0))
