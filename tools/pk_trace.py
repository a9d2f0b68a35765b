#!/usr/bin/env python3
"""Per-op timeline of the persistent flow kernel (persist.hip): every workgroup stamps s_memtime at op start / chunk done / op
complete (sts_debug_set STS_DBG_PK_TRACE).  Prints, per op, the wall time of the op (first start -> last completion over the
workgroups of an XCD, averaged over the XCDs), the longest / mean chunk compute time and the wait after the last chunk.
  python tools/pk_trace.py [phonemes] [workload]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summertts_amd import engine, synth_blob as sb   # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "hifigan_sdp"
cfg = sb.full_cfg(kind)
blob = sb.make_blob(cfg, 1234)
syn = engine.Synthesizer(blob)
ids = sb.synthetic_ids(T, cfg.vocab, salt=0)
for _ in range(3):
    syn.run_batch([ids])
syn.set_record_taps(True)
syn.debug_set("pk_trace", 1)
syn.run_batch([ids])
tr = syn.tap("pk_trace").reshape(256, 8, -1)          # [wg][start, computed, end, xcd*1000+chunk+1][step]
TICK = float(os.environ.get("PK_TICK_NS", "10.0"))      # s_memtime tick in ns (100 MHz constant clock on gfx9)
nsteps = int((tr[:, 0, :] >= 0).any(axis=0).sum())
print(f"{kind} T={T}: {nsteps} ops, {int((tr[:, 0, 0] >= 0).sum())} workgroups took part; tick = {TICK} ns")
wg_xcd = [next((int(tr[w, 3, q]) // 1000 for q in range(tr.shape[2]) if tr[w, 3, q] > 0), -1) for w in range(256)]
print("workgroups per XCD:", [wg_xcd.count(x) for x in range(8)])
for x in range(8):
    ws = [w for w in range(256) if wg_xcd[w] == x]
    if ws:
        print(f"  XCD {x}: first op start {min(tr[w, 0, 0] for w in ws) * TICK / 1e3:8.2f} us ... last op complete {max(tr[w, 2, nsteps - 1] for w in ws) * TICK / 1e3:8.2f} us")
tot = 0.0
print(" op  chunks  compute_max_us compute_mean_us  tail_wait_us | start->conv  K loop  combine  epilogue+stores  signal   (means over the workgroups that held a chunk)")
for s in range(nsteps):
    walls, cmax, cmean, tails, nch = [], [], [], [], []
    for x in range(8):
        sel = [w for w in range(256) if tr[w, 3, s] > 0 and int(tr[w, 3, s]) // 1000 == x]
        allw = [w for w in range(256) if wg_xcd[w] == x and tr[w, 0, s] >= 0 and tr[w, 2, s] >= 0]
        if not sel:
            continue
        st = np.array([tr[w, 0, s] for w in sel]); cp = np.array([tr[w, 1, s] for w in sel]); en = np.array([tr[w, 2, s] for w in sel])
        walls.append((max(tr[w, 2, s] for w in allw) - min(tr[w, 0, s] for w in allw)) * TICK / 1e3)
        cmax.append((cp - st).max() * TICK / 1e3); cmean.append((cp - st).mean() * TICK / 1e3)
        tails.append((en.max() - cp.max()) * TICK / 1e3); nch.append(len(sel))
    if walls:
        tot += np.mean(cmax) + np.mean(tails)
        sel = [w for w in range(256) if tr[w, 3, s] > 0]
        seg = ""
        if all(tr[w, 4, s] > 0 for w in sel):
            d = lambda a, b: np.mean([tr[w, b, s] - tr[w, a, s] for w in sel]) * TICK / 1e3
            seg = f" | {d(0, 4):10.2f} {d(4, 5):7.2f} {d(5, 6):8.2f} {d(6, 7):16.2f} {d(7, 1):7.2f}"
        print(f"{s:3d}  {np.mean(nch):6.1f}  {np.mean(cmax):14.2f} {np.mean(cmean):15.2f} {np.mean(tails):13.2f}{seg}")
print(f"sum of (compute_max + tail) {tot:.1f} us")
syn.close()
