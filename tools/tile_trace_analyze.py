#!/usr/bin/env python3
"""Offline analysis of a tools/tile_trace_dump.py record file (no GPU needed): per launch and group member -- workgroups, start times,
phase durations, how well the launch filled the chip's workgroup slots.  The s_memtime tick is calibrated per record from the two
real-time stamps a workgroup takes at its start and end (the shader clock moves with the load: round 3 assumed 0.45 ns, the 128-channel
launches actually run at ~0.53 ns, which made their tiles look 15 % shorter than they are).
  python tools/tile_trace_analyze.py trace.npz [--launches]"""
import sys

import numpy as np

d = np.load(sys.argv[1])
r = d["rec"]
r = r[r[:, 10] > 0]                       # complete records only
r = r[np.argsort(r[:, 3], kind="stable")]
# launches: runs of equal (gridDim, kind) whose starts are not separated by more than the previous launch's span
launches, cur = [], [0]
for i in range(1, len(r)):
    same = r[i, 0] == r[cur[0], 0] and (r[i, 2] & 0xff) == (r[cur[0], 2] & 0xff)
    if same and r[i, 3] <= r[cur, 10].max() + 100:      # started before (or within 1 us after) the end of the run so far
        cur.append(i)
    else:
        launches.append(cur); cur = [i]
launches.append(cur)
NAMES = {0: ("pro", "kloop", "epi"), 2: ("pro", "kloop", "epi"), 1: ("stage", "conv1", "park", "conv2", "epi"),      # 2 = the pre-split convs (conv_h2p.hip)
         3: ("stage", "wait_ops", "gate_k", "reduce+gate", "1x1+stores"), 4: ("stage_x0", "pre+park", "gate_k", "reduce+gate", "1x1+stores")}
print(f"{len(r)} complete records, {len(launches)} launches")
print(" #  kind  grid   wgs  span_us  slot_fill | member: n  start_us(min..max)  dur_us mean/p90  phases mean us ...")
for li, idx in enumerate(launches):
    q = r[idx]
    kind = int(q[0, 2]) & 0xff
    if kind not in NAMES:
        continue
    t0 = q[:, 3].min()
    st = (q[:, 3] - t0) * 0.01
    en = (q[:, 10] - t0) * 0.01
    last = 4 + len(NAMES[kind])
    ticks = (q[:, last] - q[:, 4]).astype(np.float64)
    tick_ns = np.median((en - st) * 1e3 / np.maximum(ticks, 1))
    span = en.max()
    # slot-time actually used / (peak concurrency x span)
    ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    peak = int(np.cumsum(ev[:, 1]).max())
    fill = (en - st).sum() / (peak * span)
    print(f"{li:2d} {('conv ', 'fused', 'h2p  ', 'flow ', 'flow0')[kind]:5s} {int(q[0, 0]):5d} {len(q):5d} {span:8.1f}  {fill:6.2f} (peak {peak} wgs, tick {tick_ns:.3f} ns)")
    for m in np.unique(q[:, 11]):
        s = q[:, 11] == m
        ph = " ".join(f"{n} {((q[s, 5 + k] - q[s, 4 + k]) * tick_ns / 1e3).mean():5.1f}" for k, n in enumerate(NAMES[kind]))
        dur = en[s] - st[s]
        print(f"        member {int(m)}: {int(s.sum()):5d}  {st[s].min():6.1f}..{st[s].max():6.1f}   {dur.mean():6.1f} / {np.percentile(dur, 90):6.1f}   {ph}")
    if "--launches" in sys.argv:
        ts = np.arange(0, span, max(span / 24, 1e-3))
        print("        running workgroups over the launch:", [int(((st <= t) & (en > t)).sum()) for t in ts])
