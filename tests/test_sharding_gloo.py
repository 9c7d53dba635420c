"""N > 1 path on CPU: two ranks over gloo exchange their keypoint counts exactly as bench.py
does over RCCL; frame sharding covers every frame once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visualslam_amd import sharding


def test_shard_range_partitions_every_frame_once():
    for n in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(sharding.shard_range(n, world, r))
            assert seen == list(range(n))
            sizes = [len(sharding.shard_range(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_gather_counts_single_process():
    c = torch.tensor([5, 9], dtype=torch.int64)
    allc = sharding.gather_counts(c)
    assert allc.tolist() == [[5, 9]]
    off, tot = sharding.global_offsets(allc, 0)
    assert off.tolist() == [0, 0] and tot.tolist() == [5, 9]
    with pytest.raises(ValueError):
        sharding.gather_counts(torch.zeros(3, dtype=torch.int64))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # each rank "detects" a rank-dependent number of keypoints on its own stream
        local = torch.tensor([100 + 10 * rank, 7 * (rank + 1)], dtype=torch.int64)
        buf = torch.zeros((world, 2), dtype=torch.int64)
        allc = sharding.gather_counts(local, buf)
        off, tot = sharding.global_offsets(allc, rank)
        frames = list(sharding.shard_range(9, world, rank))
        # the bench's max-over-ranks timing reduction
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, allc.tolist(), off.tolist(), tot.tolist(), frames, float(t)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_all = [[100, 7], [110, 14]]
    assert [r[1] for r in res] == [want_all, want_all]
    assert res[0][2] == [0, 0] and res[1][2] == [100, 7]
    assert res[0][3] == res[1][3] == [210, 21]
    assert res[0][4] + res[1][4] == list(range(9))
    assert res[0][5] == res[1][5] == 2.0
