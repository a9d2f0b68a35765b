"""Kernel micro-benchmark: the decoder / flow / text-encoder conv shapes of the full-size model through
every kernel variant (run on a GPU box).  Prints TFLOP/s (true-tap FLOPs) per variant."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summertts_amd import engine as eng

F = int(os.environ.get('CONV_BENCH_FRAMES', '668'))
SHAPES = [  # name, Cin, Cout, k, dil, L, stride_t
    ("te_qkv", 192, 576, 1, 1, 128, 0), ("te_ffn1", 192, 768, 3, 1, 128, 0), ("te_ffn2", 768, 192, 3, 1, 128, 0),
    ("flow_gate_like", 192, 384, 5, 1, F, 0), ("flow_rs", 192, 384, 1, 1, F, 0),
    ("dec_pre", 192, 512, 7, 1, F, 0), ("up1", 512, 256, 16, 1, F, 8),
    ("s1_k3", 256, 256, 3, 1, 8 * F, 0), ("s1_k7d3", 256, 256, 7, 3, 8 * F, 0), ("s1_k11d5", 256, 256, 11, 5, 8 * F, 0),
    ("up2", 256, 128, 16, 1, 8 * F, 8),
    ("s2_k3", 128, 128, 3, 1, 64 * F, 0), ("s2_k11d5", 128, 128, 11, 5, 64 * F, 0),
    ("up3", 128, 64, 4, 1, 64 * F, 2),
    ("s3_k3", 64, 64, 3, 1, 128 * F, 0), ("s3_k11d5", 64, 64, 11, 5, 128 * F, 0),
    ("up4", 64, 32, 4, 1, 128 * F, 2),
    ("mbb_s1_k3", 256, 256, 3, 1, 4 * F, 0), ("mbb_s1_k11d5", 256, 256, 11, 5, 4 * F, 0),
    ("mbb_s2_k3", 128, 128, 3, 1, 16 * F, 0), ("mbb_s2_k11d5", 128, 128, 11, 5, 16 * F, 0),
    ("s4_k3", 32, 32, 3, 1, 256 * F, 0), ("s4_k7d3", 32, 32, 7, 3, 256 * F, 0), ("s4_k11d5", 32, 32, 11, 5, 256 * F, 0),
]
if os.environ.get("CONV_BENCH_MODES"):
    _only_modes = [int(v) for v in os.environ["CONV_BENCH_MODES"].split(",")]
else:
    _only_modes = None
MODES = {50: "h2", 60: "h2_128x128", 66: "h2_4w32x128", 68: "h2s2_128x128", 74: "h2s2_4w32x128", 63: "h2_64x128", 64: "h2_32x256", 13: "bf3", 20: "bf3_128x128", 21: "bf3_64x256", 22: "bf3_128x256w8", 23: "bf3_64x128", 24: "bf3_32x256", 25: "bf3_32x128", 28: "bf3s2_128x128", 29: "bf3s2_64x256", 30: "bf3s2_128x256w8", 31: "bf3s2_64x128", 32: "bf3s2_32x256", 33: "bf3s2_32x128", 26: "bf3_4w32x128", 27: "bf3_2w32x128", 34: "bf3s2_4w32x128", 35: "bf3s2_2w32x128", 40: "kg2_128x128", 41: "pm_256x128", 42: "pm_128x128", 43: "pm_64x128", 44: "kg2_256x64", 12: "wino", 0: "auto", 2: "128x128", 3: "64x256", 4: "32x512", 5: "64x128", 6: "32x128", 7: "32x256", 8: "splitk32", 9: "splitk64"}

def main():
    only = sys.argv[1:] if len(sys.argv) > 1 else None
    rng = np.random.default_rng(0)
    for name, ci, co, k, dil, L, st in SHAPES:
        if only and name not in only:
            continue
        x = rng.standard_normal((ci, L)).astype(np.float32)
        w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        pad = (k - st) // 2 if st else dil * (k - 1) // 2
        flops = 2.0 * ci * co * k * L
        line = f"{name:14s} Cin={ci:4d} Cout={co:4d} k={k:2d} d={dil} L={L:7d} {flops/1e9:8.2f} GFLOP |"
        for mode, mname in MODES.items():
            if _only_modes is not None and mode not in _only_modes:
                continue
            if co % 64 and mode in ():
                continue
            try:
                iters = 20 if flops > 1e9 else 50
                _, ms = eng.debug_conv1d(x, w, b, pad, dil, st, False, 0.1, 1, mode=mode, iters=iters)
                line += f" {mname}:{ms*1e3:7.1f}us/{flops/ms/1e9:5.1f}TF"
            except Exception as e:
                line += f" {mname}:ERR"
        print(line, flush=True)

if __name__ == "__main__":
    main()
