#!/usr/bin/env python3
"""K-loop micro-benchmark of the pre-split trunk convs (conv_h2p.hip) against the staged two-term kernels (conv_bf3*.hip), on the REAL operand
streams: packed weights out of L2, window through LDS-DMA, the kernels' own epilogues.  Runs on a GPU box.
  python tools/ubench/kloop_h2p.py check          every tile code against the staged kernel (bit-identical) and float64
  python tools/ubench/kloop_h2p.py time           wall time per launch: shapes of the 128- and 256-channel stages x tile codes x members
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so python tools/ubench/kloop_h2p.py trace
                                                  (lab build with -DSTS_TILE_TRACE) shader cycles per (chunk, tap) step inside the K loop and the
                                                  matrix-pipe busy fraction = MFMAs x 32 cycles x waves per SIMD / cycles per step
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from summertts_amd import engine as eng   # noqa: E402

F = int(os.environ.get("KLOOP_FRAMES", "668"))
TILES = {0: ("128x128 4w 64x64", 2, 2, 4, 3), 1: ("128x128 2w 64x128", 2, 4, 2, 2), 2: ("128x128 2w 128x64", 4, 2, 2, 2), 3: ("128x256 4w 64x128", 2, 4, 4, 2),
         4: ("128x256 4w 128x64", 4, 2, 4, 2), 5: ("128x512 4w 128x128", 4, 4, 4, 1), 6: ("128x256 2w 128x128", 4, 4, 2, 1), 7: ("128x128 1w 128x128", 4, 4, 1, 1),
         8: ("64x128 1w", 2, 4, 1, 2), 9: ("64x64 1w", 2, 2, 1, 3), 10: ("128x64 1w", 4, 2, 1, 2)}     # name, MW, NW, waves / workgroup, waves / SIMD
SHAPES = [("s2_k3", 128, 3, 1, 64 * F), ("s2_k7d3", 128, 7, 3, 64 * F), ("s2_k11d5", 128, 11, 5, 64 * F),
          ("s1_k3", 256, 3, 1, 8 * F), ("s1_k7d3", 256, 7, 3, 8 * F), ("s1_k11d5", 256, 11, 5, 8 * F)]


def ref64(x, w, b, dil, res, slope_in):
    xa = np.where(x < 0, x * np.float32(slope_in), x).astype(np.float64)
    co, k, ci = w.shape
    L = x.shape[1]
    pad = dil * (k - 1) // 2
    xp = np.pad(xa, ((0, 0), (pad, pad)))
    y = np.zeros((co, L))
    for t in range(k):
        y += w[:, t, :].astype(np.float64) @ xp[:, t * dil:t * dil + L]
    y += b[:, None]
    if res is not None:
        y += res
    return y


def check():
    rng = np.random.default_rng(0)
    bad = 0
    for (C_, k, dil, L) in ((128, 3, 1, 777), (128, 7, 3, 1500), (128, 11, 5, 300), (256, 3, 1, 1029), (256, 11, 5, 97), (128, 3, 1, 5)):
        x = rng.standard_normal((C_, L)).astype(np.float32) * 1.5
        w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32)
        b = rng.standard_normal(C_).astype(np.float32)
        res = rng.standard_normal((C_, L)).astype(np.float32)
        want = eng.debug_conv1d(x, w, b, dil * (k - 1) // 2, dil, 0, False, 0.1, 1, mode=60) + res      # staged two-term kernel, 128 x 128 tile
        r64 = ref64(x, w, b, dil, res, 0.1)
        for tile in TILES:
            y, y16, yp, _ = eng.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=tile, members=2)
            lre = np.where(y < 0, y * np.float32(0.1), y)
            e_staged = np.abs(want - r64).max()          # the staged kernel's own error (its weights are packed in the natural k order, so
            e64 = np.abs(y - r64).max()                  # the matrix core sums a chunk's 16 products in another order: not bit-identical)
            e16 = np.abs(y16 - y).max()
            ep = np.abs(yp - lre).max() / max(1e-30, np.abs(lre).max())
            ok = e64 <= 1.5 * e_staged + 1e-7 and e64 < 2e-5 and e16 == 0 and ep < 2e-6
            bad += not ok
            print(f"C={C_} k={k} d={dil} L={L} tile {tile:2d}: error vs float64: staged kernel {e_staged:.2e}, pre-split {e64:.2e}  y16-y {e16:.1e}  planes rel {ep:.1e}  {'ok' if ok else 'FAIL'}", flush=True)
    print("CHECK", "FAILED" if bad else "passed")
    return bad


def time_all():
    rng = np.random.default_rng(0)
    for name, C_, k, dil, L in SHAPES:
        x = rng.standard_normal((C_, L)).astype(np.float32)
        w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32)
        b = rng.standard_normal(C_).astype(np.float32)
        res = rng.standard_normal((C_, L)).astype(np.float32)
        flops = 2.0 * C_ * C_ * k * L
        _, ms0 = eng.debug_conv1d(x, w, b, dil * (k - 1) // 2, dil, 0, False, 0.1, 1, mode=50, iters=20)
        print(f"{name:9s} {flops / 1e9:6.2f} GF | staged auto {ms0 * 1e3:6.1f} us {flops / ms0 / 1e9:6.1f} TF", flush=True)
        for members in (1, 3):
            line = f"   members={members}:"
            for tile, (tn, *_r) in TILES.items():
                try:
                    *_o, ms = eng.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=tile, members=members, iters=20)
                    line += f" t{tile}:{ms * 1e3:6.1f}us/{members * flops / ms / 1e9:5.1f}TF"
                except Exception as e:   # noqa: BLE001
                    line += f" t{tile}:ERR({e})"
            print(line, flush=True)


def trace():
    import torch
    lib = eng.load_library()
    lib.sts_debug_tile_trace.argtypes = [C.c_void_p, C.c_uint]
    WORDS, HEAD = 12, 16
    cap = 1 << 15
    buf = torch.zeros(HEAD + cap * WORDS, dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(0)
    for C_, k, dil in ((128, 3, 1), (128, 11, 5), (256, 3, 1)):
        for nwg in (1, 256, 2048):
            for tile, (tn, MW, NW, wpw, wps) in TILES.items():
                nt = int(tn.split()[0].split("x")[1])
                L = nt * nwg
                if L > 1 << 20:
                    continue
                x = rng.standard_normal((C_, L)).astype(np.float32)
                w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32)
                b = rng.standard_normal(C_).astype(np.float32)
                eng.debug_conv_h2p(x, w, b, dil, None, 0.1, 0.1, tile=tile)
                buf.zero_()
                lib.sts_debug_tile_trace(buf.data_ptr(), cap)
                eng.debug_conv_h2p(x, w, b, dil, None, 0.1, 0.1, tile=tile)
                torch.cuda.synchronize()
                n = min(lib.sts_debug_tile_trace_count(), cap)
                lib.sts_debug_tile_trace(None, 0)
                r = buf.cpu().numpy()[HEAD:].reshape(-1, WORDS)[:n]
                r = r[((r[:, 2] & 0xff) == 2) & (r[:, 7] > 0) & (r[:, 6] > 0)]
                if not len(r):
                    continue
                steps = (C_ // 16) * k
                kl = (r[:, 6] - r[:, 5]).astype(np.float64)
                ep = (r[:, 7] - r[:, 6]).astype(np.float64)
                pro = (r[:, 5] - r[:, 4]).astype(np.float64)
                wall = (r[:, 10] - r[:, 3]).astype(np.float64) * 10.0       # ns (100 MHz)
                clk = (r[:, 7] - r[:, 4]).astype(np.float64) / np.maximum(wall, 1.0)      # shader cycles per ns
                mf = 3 * MW * NW
                per = kl.mean() / steps
                # waves per SIMD actually co-resident: bounded by the launch (nwg workgroups of wpw waves on 1024 SIMDs)
                co = min(wps, max(1.0, nwg * (C_ // 128) * wpw / 1024.0))
                print(f"C={C_} k={k:2d} wgs={nwg:5d} tile {tile:2d} {tn:20s}: {per:7.0f} cycles/step ({mf} MFMAs = {mf * 32} pipe cycles; x{co:.1f} waves/SIMD -> pipe busy {mf * 32 * co / per:5.2f})  "
                      f"prologue {pro.mean():6.0f}  epilogue {ep.mean():6.0f} cycles  clock {clk.mean():.2f} GHz", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        sys.exit(1 if check() else 0)
    elif what == "time":
        time_all()
    elif what == "trace":
        trace()
