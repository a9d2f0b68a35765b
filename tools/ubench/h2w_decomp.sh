#!/bin/bash
# Winograd-domain lab kernel with parts of a step compiled out (tools/var_build.sh: VAR_SRC=conv_h2w.hip VAR_EXTRA=-DH2W_EXP=<mask> VAR_TAG=w<mask>;
# results are WRONG by design): what do the weight-fragment loads (1), the input transform + split (2) and the LDS reads (4) cost?
cd "$(dirname "$0")/../.."
for m in 0 1 2 4 7; do
  lib=summertts_amd/lib/var/libvar6w$m.so; [ $m = 0 ] && lib=summertts_amd/lib/libsummertts_hip.so
  echo "== H2W_EXP $m"
  SUMMERTTS_HIP_LIB=$lib python - <<'P' 2>&1 | grep -v amdgpu.ids
import numpy as np, sys
sys.path.insert(0, '.')
from summertts_amd import engine as eng
rng = np.random.default_rng(0)
for name, C_, k, dil, L in [("s2_k3", 128, 3, 1, 64 * 668), ("s2_k11d1", 128, 11, 1, 64 * 668), ("s2_k11d5", 128, 11, 5, 64 * 668), ("b_s2_k11d5", 128, 11, 5, 640 * 668)]:
    x = rng.standard_normal((C_, L)).astype(np.float32); w = (rng.standard_normal((C_, k, C_)) / np.sqrt(k * C_)).astype(np.float32); b = rng.standard_normal(C_).astype(np.float32)
    *_o, ms = eng.debug_conv_h2w(x, w, b, dil, None, 1.0, 1.0, members=3, iters=-10)
    *_o, ms2 = eng.debug_conv_h2w(x, w, b, dil, None, 0.1, 1.0, members=3, iters=-10)
    print(f"{name:11s} members=3: no input activation {ms * 1e3:8.1f} us   with leaky relu {ms2 * 1e3:8.1f} us")
P
done
