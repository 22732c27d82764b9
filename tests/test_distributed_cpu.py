"""world_size-2 (and 3) gloo tests of the N>1 path's host logic: shard partition + gather reassembly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genmap_amd.distributed import gather_frequency, max_shard_len, shard_ranges


def test_shard_ranges_cover_and_align():
    for nk, step, world in ((1000, 22, 2), (1000, 22, 8), (5, 7, 4), (248956393, 22, 8), (100, 1, 3), (0, 5, 2)):
        r = shard_ranges(nk, step, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == nk
        for (a, b), (c, d) in zip(r[:-1], r[1:]):
            assert b == c and a <= b
        for a, b in r:
            assert a % step == 0 or a == nk


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nk, step, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges = shard_ranges(nk, step, world)
    m = max_shard_len(ranges)
    truth = (torch.arange(nk + 29, dtype=torch.int64) * 7 % 251).to(torch.uint8)
    truth[nk:] = 0
    local = torch.zeros(nk + 29 + m, dtype=torch.uint8)
    b, e = ranges[rank]
    local[b:e] = truth[b:e]            # what this rank's gm_map_device(kmer_range=(b, e)) would have written
    gather_frequency(local, ranges, rank, world, dist)
    if rank == 0:
        q.put(bool(torch.equal(local[:nk + 29], truth)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_reassembles_frequency_vector(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 10007, 22, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
    assert ok
