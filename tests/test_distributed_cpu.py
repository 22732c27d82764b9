"""world_size-2 (and 3) gloo tests of the N>1 path's host logic: shard partition + gather reassembly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genmap_amd.distributed import PiecePlan, ShardPlan, gather_chunks, gather_frequency, max_shard_len, shard_ranges


def test_shard_ranges_cover_and_align():
    for nk, step, world in ((1000, 22, 2), (1000, 22, 8), (5, 7, 4), (248956393, 22, 8), (100, 1, 3), (0, 5, 2)):
        r = shard_ranges(nk, step, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == nk
        for (a, b), (c, d) in zip(r[:-1], r[1:]):
            assert b == c and a <= b
        for a, b in r:
            assert a % step == 0 or a == nk


def test_shard_plan_partitions_every_position_once():
    """interleaved chunks: every k-mer position belongs to exactly one rank, chunks are whole blocks, and every rank
    gets its share of every region (>= 50 chunks per rank on a large text, src/algo.hpp:422-428)"""
    for nk, step, world in ((1000, 22, 2), (1000, 7, 3), (5, 7, 4), (248956393, 15, 8), (100, 1, 3), (0, 5, 2), (3088269803, 5, 8)):
        plan = ShardPlan(nk, step, world)
        assert plan.rows % world == 0 and plan.rows * plan.chunk_len >= nk
        assert plan.chunk_len % step == 0
        total = 0
        prev_end = {}
        seen = []
        for r in range(world):
            for b, e in plan.positions_of(r):
                assert b % step == 0 and b < e <= nk
                seen.append((b, e, r))
                total += e - b
        assert total == nk
        seen.sort()
        for (b0, e0, _), (b1, e1, _) in zip(seen[:-1], seen[1:]):
            assert e0 == b1
        if nk > 10**8:
            for r in range(world):
                assert len(plan.positions_of(r)) >= 50
        assert plan.chunk_arg(0) is None if world == 1 else plan.chunk_arg(1) == (plan.chunk_blocks, 1, world)


def test_piece_plan_carries_every_chunk_into_separately_allocated_pieces():
    """PeerGather's arithmetic without a GPU: the root's vector is several allocations of whole chunk rows (an IPC handle of one
    3 GB allocation could not be opened); every rank's strided copies of every launch, replayed on numpy arrays, must leave the
    pieces equal to the unsharded vector -- for piece sizes from one row to everything, 8- and 16-bit items, ragged last chunks"""
    for nk, step, world, cpr, K in ((100_003, 7, 2, 5, 24), (99_991, 15, 3, 4, 30), (5_000, 1, 8, 3, 4), (64, 7, 4, 2, 3), (30_000, 5, 1, 6, 30)):
        plan = ShardPlan(nk, step, world, chunks_per_rank=cpr)
        n = nk + K - 1
        for item in (1, 2):
            truth = ((np.arange(n * item, dtype=np.int64) * 131 + 7) % 251).astype(np.uint8)
            truth[nk * item:] = 0                                          # the zeros of resetLimits behind the last k-mer
            for piece_bytes in (1, 3 * plan.chunk_len * world * item, 1 << 40):
                pp = PiecePlan(plan, n, item, piece_bytes)
                assert pp.piece_off[0] == 0 and sum(pp.piece_len) == pp.nbytes == plan.padded_len(n) * item
                assert all(o % pp.row_bytes == 0 for o in pp.piece_off)
                if piece_bytes == 1:
                    assert len(pp.piece_off) == max(1, plan.rows // world)
                pieces = [np.zeros(m, np.uint8) for m in pp.piece_len]      # gm_device_alloc zeroes
                for rank in range(world):
                    local = np.zeros(pp.nbytes, np.uint8)
                    for b, e in plan.positions_of(rank):                    # what the rank's kernel writes: its own chunks ...
                        local[b * item:e * item] = truth[b * item:e * item]
                    # (... and the zeros at the end of every sequence, which leave `local` as it is)
                    for sub in plan.sub_ranges(3):
                        for q, in_piece, in_vector, pitch, nbytes, count in pp.copies(rank, sub):
                            assert 0 <= in_piece and in_vector - in_piece == pp.piece_off[q]
                            for i in range(count):
                                assert in_piece + i * pitch + nbytes <= pp.piece_len[q], "a chunk must not straddle pieces"
                                pieces[q][in_piece + i * pitch:in_piece + i * pitch + nbytes] = local[in_vector + i * pitch:in_vector + i * pitch + nbytes]
                got = np.concatenate(pieces)[:n * item]
                assert np.array_equal(got, truth), (nk, step, world, item, piece_bytes)
                for off in (0, pp.nbytes - 1, pp.nbytes // 2):
                    q, o = pp.locate(off)
                    assert pp.piece_off[q] + o == off and 0 <= o < pp.piece_len[q]


def _chunk_worker(rank, world, port, nk, step, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = ShardPlan(nk, step, world, chunks_per_rank=5)
    n = nk + 29
    ok = True
    for dt, mod in ((torch.uint8, 251), (torch.uint16, 65521)):
        truth = torch.from_numpy(((np.arange(n, dtype=np.int64) * 977) % mod).astype(np.uint8 if dt == torch.uint8 else np.uint16))
        truth[nk:] = 0
        local = torch.zeros(plan.padded_len(n), dtype=dt)
        for b, e in plan.positions_of(rank):     # what this rank's gm_map_device(chunks=plan.chunk_arg(rank)) would have written
            local[b:e] = truth[b:e]
        gather_chunks(local, plan, rank, dist)
        if rank == 0:
            ok &= bool(torch.equal(local[:n].view(torch.uint8), truth.view(torch.uint8)))
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_chunks_reassembles_frequency_vector(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, 10007, 22, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
    assert ok


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
    # 16-bit frequencies (-fl / mappability): the same exchange as raw bytes
    truth16 = torch.from_numpy(((np.arange(nk + 29, dtype=np.int64) * 977) % 65521).astype(np.uint16))
    truth16[nk:] = 0
    local16 = torch.zeros(nk + 29 + m, dtype=torch.uint16)
    local16[b:e] = truth16[b:e]
    gather_frequency(local16, ranges, rank, world, dist)
    if rank == 0:
        q.put(bool(torch.equal(local[:nk + 29], truth)) and bool(torch.equal(local16[:nk + 29].view(torch.int16), truth16.view(torch.int16))))
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


def _fake_locations(begin, npos, seed):
    rng = np.random.default_rng(seed)
    cp, cm = rng.integers(0, 4, size=npos), rng.integers(0, 3, size=npos)
    po = np.concatenate([[0], np.cumsum(cp)]).astype(np.uint64); mo = np.concatenate([[0], np.cumsum(cm)]).astype(np.uint64)
    pl = rng.integers(0, 2**40, size=int(po[-1]), dtype=np.uint64); mi = rng.integers(0, 2**40, size=int(mo[-1]), dtype=np.uint64)
    return begin, po, pl, mo, mi


def _loc_worker(rank, world, port, q):
    from genmap_amd.distributed import gather_locations
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges = shard_ranges(3001, 6, world)
    ranges[-1] = (ranges[-1][0], ranges[-1][0]) if world == 3 else ranges[-1]     # world 3: the last shard is empty
    shards = [_fake_locations(b, e - b, 100 + r) for r, (b, e) in enumerate(ranges)]
    got = gather_locations(shards[rank], rank, world, dist)
    if rank == 0:
        pb, po, pl, mo, mi = got
        ok = pb == 0 and len(po) == sum(len(s[1]) - 1 for s in shards) + 1
        j = 0
        for s in shards:                      # every position's two lists survive the exchange
            for i in range(len(s[1]) - 1):
                ok &= np.array_equal(pl[int(po[j]):int(po[j + 1])], s[2][int(s[1][i]):int(s[1][i + 1])])
                ok &= np.array_equal(mi[int(mo[j]):int(mo[j + 1])], s[4][int(s[3][i]):int(s[3][i + 1])])
                j += 1
        q.put(bool(ok))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_locations_counts_first_then_payload(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
    assert ok
