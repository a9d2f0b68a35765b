"""Wall time of the HiFi-GAN generator's non-ResBlock convs at one utterance (668 frames), per tile code of the two-term kernels.
  python tools/ubench/ups_tiles.py            (GPU box)
Shapes: conv_pre 192->512 k7; ups 512->256 k16 s8, 256->128 k16 s8, 128->64 k4 s2, 64->32 k4 s2 (reference Generator_hifigan.cpp:60-118)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from summertts_amd import engine

CASES = [("conv_pre", 192, 512, 7, 3, 668, 0), ("ups0", 512, 256, 16, 4, 668, 8), ("ups1", 256, 128, 16, 4, 5344, 8),
         ("ups2", 128, 64, 4, 1, 42752, 2), ("ups3", 64, 32, 4, 1, 85504, 2)]
MODES = [0, 13, 50, 60, 63, 64, 80, 82, 83, 150, 160, 163, 180, 182, 183]   # + 100: row-interleaved phases (ConvArgs::rowph) + ([61, 62, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 81] if engine.lab_build() else [])
rng = np.random.default_rng(0)
for name, ci, co, k, pad, L, st in CASES:
    x = rng.standard_normal((ci, L)).astype(np.float32)
    w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    row = []
    for m in MODES:
        if m >= 100 and not st:
            continue
        try:
            _, ms = engine.debug_conv1d(x, w, b, pad, 1, st, False, in_slope=0.1, in_act=1, mode=m, iters=200)
            row.append("%d:%.1f" % (m, ms * 1e3))
        except Exception as e:
            row.append("%d:-" % m)
    print("%-9s us per launch by mode  %s" % (name, "  ".join(row)), flush=True)
