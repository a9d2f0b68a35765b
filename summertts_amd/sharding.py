"""Multi-GPU utterance sharding (new capability; the reference synthesises one utterance per call).

Utterances are independent (no cross-utterance state, /root/reference/src/models/SynthesizerTrn.cpp:323),
so the batch is sharded by utterance with NO collective on the data path; weights are replicated.
The only exchange is the final gather of the variable-length int16 PCM to rank 0 -- one
``all_gather`` of the sample counts and one ``gather`` of a max-padded int16 buffer over RCCL/xGMI
(payload <= tens of MB per step, so a direct gather into root over its 7 point-to-point links is
enough; no ring, no all-reduce anywhere).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_utterances(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first greedy bin packing of utterance indices into ``world`` shards balanced by the
    phoneme count (work is ~proportional to it).  Deterministic; indices inside a shard are ascending."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], len(shards[k]), k))
        shards[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(s) for s in shards]


def gather_variable(local, counts_local, dist, torch, rank: int, world: int, max_utts: int):
    """Gathers one variable-length int16 tensor per rank to rank 0.

    local: 1-D int16 tensor (on the backend's device) holding this rank's utterances back to back;
    counts_local: per-utterance sample counts (python ints).  Returns on rank 0 a list (per rank) of
    lists (per utterance) of numpy int16 arrays; ``None`` elsewhere."""
    dev = local.device
    cnt = torch.zeros(max_utts + 1, dtype=torch.int64, device=dev)
    cnt[0] = len(counts_local)
    if len(counts_local):
        cnt[1:1 + len(counts_local)] = torch.as_tensor(list(counts_local), dtype=torch.int64, device=dev)
    all_cnt = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(all_cnt, cnt)
    totals = [int(c[1:1 + int(c[0])].sum().item()) for c in all_cnt]
    cap = max(1, max(totals))
    buf = torch.zeros(cap, dtype=torch.int16, device=dev)
    buf[:local.numel()] = local
    # int16 travels as raw bytes: every backend (RCCL, gloo) moves uint8
    bbuf = buf.view(torch.uint8)
    gl = [torch.zeros(cap * 2, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(bbuf, gl, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        host = gl[r].view(torch.int16).cpu().numpy()
        k = int(all_cnt[r][0].item())
        cs = [int(v) for v in all_cnt[r][1:1 + k].tolist()]
        offs = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
        out.append([host[offs[i]:offs[i + 1]].copy() for i in range(k)])
    return out


def gather_pcm(syn, n_out, dist, torch, rank: int, world: int, max_utts: int = 0):
    """Device-side PCM of the last ``syn.run_batch`` -> rank 0 (RCCL).  Returns (per-rank lists, counts)."""
    total = int(np.asarray(n_out).sum())
    local = torch.empty(max(1, total), dtype=torch.int16, device="cuda")
    syn.pcm_to_device_ptr(local.data_ptr(), local.numel())
    res = gather_variable(local[:total], [int(v) for v in n_out], dist, torch, rank, world,
                          max_utts or max(1, len(n_out)))
    return res, n_out
