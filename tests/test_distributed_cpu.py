"""The N > 1 path (image sharding + statistics all-reduce) rehearsed with 2 gloo ranks on the CPU."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from compression_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = D.StatsAccumulator()
    for i in D.shard_indices(n_images, rank, world):
        acc.add_image(bits=1000.0 * (i + 1), pixels=64 * 64, sq_err=0.5 * (i + 1), nbytes=125 * (i + 1), seconds=0.01)
    res = acc.all_reduce().result()
    if rank == 0:
        out.put(res)
    dist.destroy_process_group()


def test_shards_are_a_partition():
    for n, w in ((24, 8), (5, 2), (3, 4), (0, 2)):
        seen = sorted(i for r in range(w) for i in D.shard_indices(n, r, w))
        assert seen == list(range(n))


def test_two_rank_stats_allreduce_matches_single_process():
    n_images, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = D.StatsAccumulator()
    for i in range(n_images):
        single.add_image(bits=1000.0 * (i + 1), pixels=64 * 64, sq_err=0.5 * (i + 1), nbytes=125 * (i + 1), seconds=0.01)
    want = single.result()
    for k in want:
        assert res[k] == pytest.approx(want[k], rel=1e-12), k
    assert res["images"] == n_images
