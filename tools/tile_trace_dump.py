#!/usr/bin/env python3
"""Raw per-workgroup phase records of the staged / fused trunk kernels inside one real synthesis step (lab build with -DSTS_TILE_TRACE:
VAR_EXTRA=-DSTS_TILE_TRACE VAR_TAG=tt tools/var_build.sh 6), written to an .npz for offline analysis (tools/tile_trace_analyze.py runs
without a GPU).  Record (12 words, conv_bf3_dev.hpp): gridDim.x, blockIdx.x, kind | HW_ID << 8 | XCC_ID << 40, realtime (100 MHz) at
start, six s_memtime stamps, realtime at end, group member.
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so python tools/tile_trace_dump.py out.npz [batch] [conv_math]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summertts_amd import engine, synth_blob as sb   # noqa: E402

WORDS, HEAD = 12, 16
out = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
math = sys.argv[3] if len(sys.argv) > 3 else None
cfg = sb.full_cfg(os.environ.get("TT_WORKLOAD", "hifigan_sdp"))
blob = sb.make_blob(cfg, 1234)
syn = engine.Synthesizer(blob)
if math:
    syn.set_conv_math(math)
lens = [128] if B == 1 else np.random.default_rng(1234).integers(64, 257, size=B).tolist()
ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]
for _ in range(3):
    syn.run_batch(ids)
lib = syn.lib
cap = 1 << 18
buf = torch.zeros(HEAD + cap * WORDS, dtype=torch.int64, device="cuda")
lib.sts_debug_tile_trace.argtypes = [C.c_void_p, C.c_uint]
assert lib.sts_debug_tile_trace(buf.data_ptr(), cap) == 0
syn.run_batch(ids)
torch.cuda.synchronize()
n = lib.sts_debug_tile_trace_count()
lib.sts_debug_tile_trace(None, 0)
r = buf.cpu().numpy()[HEAD:].reshape(-1, WORDS)[:min(n, cap)]
np.savez_compressed(out, rec=r, batch=B, lens=np.asarray(lens))
print(f"{n} workgroup records -> {out}")
