#!/usr/bin/env python3
"""Phase timeline of the staged split-bf16 conv kernel inside a real synthesis step (lab build: tools/var_build.sh with
VAR_EXTRA=-DSTS_TILE_TRACE).  Every workgroup records s_memtime at start / first barrier (prologue done) / K loop done /
epilogue stores complete, plus the chip-wide 100 MHz real-time counter at its start.  Prints, per launch (grouped by grid size
and start time): workgroups, the phases' mean / p90 durations, the launch's span, and how many workgroups started within the
first 2 us (co-resident wave) -- i.e. how much of a tile's life is not the K loop, and whether the tiles run in lockstep.
  VAR_EXTRA=-DSTS_TILE_TRACE tools/var_build.sh 6 && SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6.so python tools/tile_trace.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summertts_amd import engine, synth_blob as sb   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TICK = float(os.environ.get("TT_TICK_NS", "0.45"))
cfg = sb.full_cfg("hifigan_sdp")
blob = sb.make_blob(cfg, 1234)
syn = engine.Synthesizer(blob)
lens = [128] if B == 1 else np.random.default_rng(1234).integers(64, 257, size=B).tolist()
ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]
if os.environ.get("TT_TRUNK_MODE"):
    syn.debug_set("trunk_mode", int(os.environ["TT_TRUNK_MODE"]))
for _ in range(3):
    syn.run_batch(ids)
lib = syn.lib
cap = 1 << 17
buf = torch.zeros(cap * 10, dtype=torch.int64, device="cuda")
lib.sts_debug_tile_trace.argtypes = [C.c_void_p, C.c_uint]
assert lib.sts_debug_tile_trace(buf.data_ptr(), cap) == 0
syn.run_batch(ids)
torch.cuda.synchronize()
n = lib.sts_debug_tile_trace_count()
lib.sts_debug_tile_trace(None, 0)
r = buf.cpu().numpy().reshape(-1, 10)[:min(n, cap)]
print(f"{n} workgroup records, batch {B}; s_memtime tick assumed {TICK} ns (shader clock), real-time tick 10 ns")
order = np.argsort(r[:, 3], kind="stable")
r = r[order]
# launches: runs of equal grid size in start-time order
launches, cur = [], [0]
for i in range(1, len(r)):
    if r[i, 0] != r[cur[0], 0] or (r[i, 3] - r[cur[-1], 3]) * 10e-3 > 30.0:
        launches.append(cur); cur = [i]
    else:
        cur.append(i)
launches.append(cur)
print(" launch kind   wgs  span_us | phases (mean us, p90 in brackets)                                                       | non-MFMA share | started in first 2 us")
for li, idx in enumerate(launches):
    q = r[idx]
    kind = int(q[0, 2]) & 0xff
    last = 9 if kind == 1 else 7
    ok = (q[:, last] > 0) & (q[:, 5] > 0)
    if not ok.any():
        continue
    q = q[ok]
    t0 = (q[:, 3] - q[:, 3].min()) * 10e-3
    end = t0 + (q[:, last] - q[:, 4]) * TICK / 1e3
    first = int((t0 < 2.0).sum())
    d = lambda a, b: (q[:, b] - q[:, a]) * TICK / 1e3
    f = lambda v: f"{v.mean():6.2f} ({np.percentile(v, 90):6.2f})"
    if kind == 0:
        pro, kl, ep = d(4, 5), d(5, 6), d(6, 7)
        ph = f"prologue {f(pro)}  K loop {f(kl)}  epilogue {f(ep)}"
        share = 100 * (pro.sum() + ep.sum()) / (pro.sum() + kl.sum() + ep.sum())
    else:
        st, c1, pk, c2, ep = d(4, 5), d(5, 6), d(6, 7), d(7, 8), d(8, 9)
        ph = f"stage {f(st)}  conv1 {f(c1)}  park {f(pk)}  conv2 {f(c2)}  epilogue {f(ep)}"
        share = 100 * (st.sum() + pk.sum() + ep.sum()) / (st.sum() + c1.sum() + pk.sum() + c2.sum() + ep.sum())
    print(f"{li:7d} {'fused' if kind else 'conv '} {len(q):5d} {end.max():8.1f} | {ph:110s} | {share:9.1f} % | {first}")
# persistent stage kernel: per workgroup lifetime, time spent waiting for dependencies, items (records of kind 2)
k2 = r[(r[:, 2] & 0xff) == 2]
if len(k2):
    life = (k2[:, 5] - k2[:, 4]) * TICK / 1e3; wait = k2[:, 6] * TICK / 1e3
    xc = (k2[:, 2] >> 40) & 0xf
    print(f"stage kernel: {len(k2)} workgroups, lifetime mean {life.mean():.1f} us (max {life.max():.1f}), dependency wait mean {wait.mean():.1f} us "
          f"({100 * wait.sum() / life.sum():.1f} % of lifetime), items per workgroup mean {k2[:, 7].mean():.1f} (min {k2[:, 7].min()}, max {k2[:, 7].max()})")
    for x in range(8):
        sel = xc == x
        if sel.any():
            print(f"  XCD {x}: {int(sel.sum())} workgroups, lifetime {life[sel].mean():7.1f} us, wait {wait[sel].mean():6.1f} us, items {k2[sel, 7].sum()}")
r = r[(r[:, 2] & 0xff) != 2]
# placement: which workgroups share a CU?  (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC id on top)
if os.environ.get("TT_PLACEMENT"):
    li = int(os.environ["TT_PLACEMENT"])
    q = r[launches[li]]
    hw = (q[:, 2] >> 8) & 0xffffffff
    xcc = (q[:, 2] >> 40) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    t0 = (q[:, 3] - q[:, 3].min()) * 10e-3
    print(f"placement of launch {li}: {len(np.unique(cu))} distinct CU ids")
    by = {}
    for blk, c, tt, kl in zip(q[:, 1], cu, t0, (q[:, 6] - q[:, 5]) * TICK / 1e3):
        by.setdefault(int(c), []).append((int(blk), round(float(tt), 1), round(float(kl), 1)))
    for c in sorted(by)[:48]:
        print(f"  cu {c:5d} (xcc {c >> 8}): " + "  ".join(f"blk {b_:4d} t0 {t_:5.1f} K {k_:5.1f}" for b_, t_, k_ in sorted(by[c], key=lambda z: z[1])))
syn.close()
