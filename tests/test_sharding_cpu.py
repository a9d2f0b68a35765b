"""CPU suite: utterance sharding and the variable-length PCM gather (gloo, world_size 2)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from summertts_amd import sharding


def test_shard_utterances_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    lens = rng.integers(64, 257, size=256).tolist()
    for world in (1, 2, 4, 8):
        shards = sharding.shard_utterances(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(256))
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)
        assert all(s == sorted(s) for s in shards)
    assert sharding.shard_utterances([5, 5, 5], 4) == [[0], [1], [2], []]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [100, 7, 300, 41, 13]
    shards = sharding.shard_utterances(lens, world)
    mine = shards[rank]
    max_utts = max(len(s) for s in shards)
    pcs = [np.full(lens[u] * 3, u + 1, np.int16) + np.arange(lens[u] * 3, dtype=np.int16) for u in mine]
    local = torch.from_numpy(np.concatenate(pcs)) if pcs else torch.zeros(0, dtype=torch.int16)
    res = sharding.gather_variable(local, [p.size for p in pcs], dist, torch, rank, world, max_utts)
    # pipelined form used by bench.py: the gather of step k is still in flight while step k + 1 is issued
    local2 = (local + 7).to(torch.int16)
    h1 = sharding.begin_gather(local, [p.size for p in pcs], dist, torch, rank, world, max_utts)
    h2 = sharding.begin_gather(local2, [p.size for p in pcs], dist, torch, rank, world, max_utts)
    res1 = sharding.finish_gather(h1)
    res2 = sharding.finish_gather(h2)
    if rank == 0:
        ok = True
        for r in range(world):
            for k, u in enumerate(shards[r]):
                exp = np.full(lens[u] * 3, u + 1, np.int16) + np.arange(lens[u] * 3, dtype=np.int16)
                ok = ok and np.array_equal(res[r][k], exp) and np.array_equal(res1[r][k], exp) and np.array_equal(res2[r][k], exp + 7)
        q.put(ok)
    else:
        assert res is None and res1 is None and res2 is None
    # shards of different sizes, one of them EMPTY, and max_utts left to the collective (all_reduce MAX): every rank
    # must still build count tensors of the same length and join both collectives
    lens3 = [9]
    sh3 = sharding.shard_utterances(lens3, world)
    assert [len(x) for x in sh3] == [1, 0]
    pcs3 = [np.arange(lens3[u] * 2, dtype=np.int16) + 5 for u in sh3[rank]]
    loc3 = torch.from_numpy(np.concatenate(pcs3)) if pcs3 else torch.zeros(0, dtype=torch.int16)
    res3 = sharding.gather_variable(loc3, [p.size for p in pcs3], dist, torch, rank, world)
    if rank == 0:
        q.put(len(res3) == 2 and len(res3[1]) == 0 and np.array_equal(res3[0][0], np.arange(18, dtype=np.int16) + 5))
    assert sharding.run_shard(None, [], [], []).size == 0          # an empty shard never touches the engine
    dist.destroy_process_group()


def test_gather_variable_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
    assert q.get(timeout=5) is True
