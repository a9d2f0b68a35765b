#!/usr/bin/env python3
"""K-loop time per 16-channel step of ONE staged split-bf16 conv (lab build with -DSTS_TILE_TRACE) as a function of how many
workgroups share the chip: 1 workgroup (nothing to contend with), one per CU, two per CU, many.  128 -> 128 channels, 128 x 128 tile.
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6.so python tools/tile_trace_conv.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summertts_amd import engine   # noqa: E402

TICK = float(os.environ.get("TT_TICK_NS", "0.45"))
lib = engine.load_library()
lib.sts_debug_tile_trace.argtypes = [C.c_void_p, C.c_uint]
cap = 1 << 15
buf = torch.zeros(cap * 10, dtype=torch.int64, device="cuda")
rng = np.random.default_rng(0)
for k, dil in ((3, 1), (11, 5)):
    for mode, mname in (((60, "f16x2 128x128 (4 waves of 64x64)"), (20, "bf16x3 128x128 (4 waves of 64x64)")) if os.environ.get("TT_H2") else
                        ((20, "128x128 (4 waves of 64x64)"), (26, "128x128 (4 waves of 32x128)"))):
        for nwg in (1, 64, 256, 512, 1024, 4096):
            L = 128 * nwg
            x = rng.standard_normal((128, L)).astype(np.float32)
            w = (rng.standard_normal((128, k, 128)) / np.sqrt(k * 128)).astype(np.float32)
            b = rng.standard_normal(128).astype(np.float32)
            engine.debug_conv1d(x, w, b, dil * (k - 1) // 2, dil, 0, False, 0.1, 1, mode=mode)       # warm-up
            buf.zero_()
            lib.sts_debug_tile_trace(buf.data_ptr(), cap)
            engine.debug_conv1d(x, w, b, dil * (k - 1) // 2, dil, 0, False, 0.1, 1, mode=mode)
            torch.cuda.synchronize()
            n = min(lib.sts_debug_tile_trace_count(), cap)
            lib.sts_debug_tile_trace(None, 0)
            r = buf.cpu().numpy().reshape(-1, 10)[:n]
            r = r[(r[:, 7] > 0) & (r[:, 5] > 0)]
            steps = 8 * k
            kl = (r[:, 6] - r[:, 5]) * TICK / 1e3
            pro = (r[:, 5] - r[:, 4]) * TICK / 1e3
            ep = (r[:, 7] - r[:, 6]) * TICK / 1e3
            print(f"k={k:2d} {mname:36s} {nwg:5d} workgroups: K loop {kl.mean():7.2f} us = {1e3 * kl.mean() / steps / TICK:6.0f} ticks per step "
                  f"({12 if mode >= 60 else 24} MFMAs = {384 if mode >= 60 else 768} pipe cycles)   prologue {pro.mean():5.2f}  epilogue {ep.mean():5.2f} us", flush=True)
