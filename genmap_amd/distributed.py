"""One-process-per-GPU sharding of a computeMappability call and the gather of its result.

Every rank holds a full index replica (2 BWTs ~ 1.3 B/symbol each: a 3.1 Gbp index is ~4 GB of 288 GB) and
computes a disjoint, contiguous range of whole k-mer blocks; positions are independent given the read-only
index (the only coupling in the reference, copying a value to duplicate k-mers at src/algo.hpp:389-396, is
an optimisation the GPU build drops).  The only collective is ONE gather of the 8/16-bit frequency shards to
the root over RCCL/xGMI -- no reduction, positions are disjoint.  Works with any torch.distributed backend
(nccl = RCCL on ROCm; gloo in the CPU tests)."""
from typing import List, Tuple


def shard_ranges(num_kmers: int, step_size: int, world: int) -> List[Tuple[int, int]]:
    """[kmer_begin, kmer_end) per rank: whole blocks of step_size k-mers, contiguous, covering [0, num_kmers)."""
    nblocks = (num_kmers + step_size - 1) // step_size
    out = []
    for r in range(world):
        b0, b1 = nblocks * r // world, nblocks * (r + 1) // world
        out.append((min(b0 * step_size, num_kmers), min(b1 * step_size, num_kmers)))
    return out


def max_shard_len(ranges) -> int:
    return max((e - b) for b, e in ranges) if ranges else 0


def gather_frequency(local_full, ranges, rank: int, world: int, dist, recv_bufs=None, dst: int = 0, stage_on_host: bool = False):
    """local_full: this rank's frequency vector (1-D tensor with at least ranges[-1][1] + max shard elements,
    own shard filled, zeros elsewhere).  After the call the dst rank's local_full holds every rank's shard.
    recv_bufs: optional preallocated list of `world` tensors of max_shard_len elements on dst."""
    if world == 1:
        return local_full
    m = max_shard_len(ranges)
    b, _ = ranges[rank]
    send = local_full[b:b + m]
    if stage_on_host:   # backends without device-memory collectives (gloo rehearsal): same data path through host buffers
        send = send.cpu()
        recv_bufs = None
    if rank == dst:
        if recv_bufs is None:
            recv_bufs = [send.new_empty(m) for _ in range(world)]
        dist.gather(send, recv_bufs, dst=dst)
        for r, (rb, re) in enumerate(ranges):
            if r != dst and re > rb:
                local_full[rb:re] = recv_bufs[r][:re - rb].to(local_full.device)
    else:
        dist.gather(send, None, dst=dst)
    return local_full
