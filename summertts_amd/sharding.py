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


class PendingGather:
    """Handle of a gather whose payload is still in flight (see begin_gather / finish_gather)."""
    __slots__ = ("work", "big", "buf", "cnts", "rank", "world")


def begin_gather(local, counts_local, dist, torch, rank: int, world: int, max_utts: int = 0) -> PendingGather:
    """Starts the gather of one variable-length int16 tensor per rank to rank 0 and returns at once.

    local: 1-D int16 tensor (on the backend's device) holding this rank's utterances back to back (empty for a rank
    whose shard is empty -- it still takes part in both collectives);
    counts_local: per-utterance sample counts (python ints).  The sample counts are exchanged first (one tiny
    all_gather + one download: the payload buffers are sized from them); the PCM itself travels asynchronously
    (``async_op``) so that the caller can start the next batch while RCCL moves it.
    max_utts: the largest shard size over ALL ranks (shards differ in size: ``shard_utterances([5, 5, 5], 4)`` leaves
    rank 3 empty).  0 = agree on it here with one all_reduce(MAX), so ranks can never build count tensors of
    different lengths."""
    dev = local.device
    if max_utts <= 0:
        mx = torch.tensor([len(counts_local)], dtype=torch.int64, device=dev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        max_utts = int(mx.item())
    max_utts = max(1, max_utts)
    if len(counts_local) > max_utts:
        raise ValueError(f"shard of {len(counts_local)} utterances exceeds max_utts={max_utts}")
    cnt_h = np.zeros(max_utts + 1, dtype=np.int64)
    cnt_h[0] = len(counts_local)
    cnt_h[1:1 + len(counts_local)] = list(counts_local)
    cnt = torch.from_numpy(cnt_h).to(dev)
    all_cnt = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(all_cnt, cnt)
    cnts = torch.stack(all_cnt).cpu().numpy()
    totals = [int(cnts[r, 1:1 + int(cnts[r, 0])].sum()) for r in range(world)]
    cap = max(1, max(totals))
    # int16 travels as raw bytes: every backend (RCCL, gloo) moves uint8.  The tail beyond this rank's samples
    # is never read on the root, so the send buffer is not cleared.
    buf = torch.empty(cap, dtype=torch.int16, device=dev)
    buf[:local.numel()] = local
    h = PendingGather()
    h.buf = buf; h.cnts = cnts; h.rank = rank; h.world = world
    h.big = torch.empty((world, cap * 2), dtype=torch.uint8, device=dev) if rank == 0 else None
    h.work = dist.gather(buf.view(torch.uint8), list(h.big.unbind(0)) if rank == 0 else None, dst=0, async_op=True)
    return h


def finish_gather(h: PendingGather):
    """Completes begin_gather: on rank 0 a list (per rank) of lists (per utterance) of numpy int16 arrays,
    ``None`` elsewhere."""
    h.work.wait()
    if h.rank != 0:
        return None
    host = h.big.cpu().numpy().view(np.int16)      # [world, cap]: one device-to-host copy
    out = []
    for r in range(h.world):
        k = int(h.cnts[r, 0])
        cs = [int(v) for v in h.cnts[r, 1:1 + k]]
        offs = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
        out.append([host[r, offs[i]:offs[i + 1]].copy() for i in range(k)])
    return out


def gather_variable(local, counts_local, dist, torch, rank: int, world: int, max_utts: int = 0):
    """Blocking form: begin_gather + finish_gather."""
    return finish_gather(begin_gather(local, counts_local, dist, torch, rank, world, max_utts))


def _local_pcm(syn, n_out, torch):
    """This rank's PCM of the last ``syn.run_batch`` as a device tensor (empty shard: no run happened, nothing to copy)."""
    total = int(np.asarray(n_out).sum()) if len(n_out) else 0
    local = torch.empty(max(1, total), dtype=torch.int16, device="cuda")
    if len(n_out):
        syn.pcm_to_device_ptr(local.data_ptr(), local.numel())
    return local[:total]


def run_shard(syn, ids, sid, length_scale):
    """``syn.run_batch`` that tolerates an empty shard (a rank with no utterances skips the engine but must still
    join the gather collectives)."""
    if len(ids) == 0:
        return np.zeros(0, np.int32)
    return syn.run_batch(ids, sid, length_scale)


def gather_pcm(syn, n_out, dist, torch, rank: int, world: int, max_utts: int = 0):
    """Device-side PCM of the last ``syn.run_batch`` -> rank 0 (RCCL).  Returns (per-rank lists, counts).
    max_utts = 0: agreed on with an all_reduce(MAX) inside begin_gather."""
    res = gather_variable(_local_pcm(syn, n_out, torch), [int(v) for v in n_out], dist, torch, rank, world, max_utts)
    return res, n_out


def begin_gather_pcm(syn, n_out, dist, torch, rank: int, world: int, max_utts: int = 0) -> PendingGather:
    """Asynchronous form of gather_pcm: the engine's PCM is copied out (device to device) and handed to RCCL;
    the engine is free for the next batch as soon as this returns."""
    return begin_gather(_local_pcm(syn, n_out, torch), [int(v) for v in n_out], dist, torch, rank, world, max_utts)
