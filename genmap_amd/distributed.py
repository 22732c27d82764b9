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
    import torch
    item = local_full.element_size()
    send = local_full[b:b + m].view(torch.uint8)   # raw bytes: collectives do not take 16-bit unsigned elements
    if stage_on_host:   # backends without device-memory collectives (gloo rehearsal): same data path through host buffers
        send = send.cpu()
        recv_bufs = None
    if rank == dst:
        if recv_bufs is None:
            recv_bufs = [send.new_empty(m * item) for _ in range(world)]
        else:
            recv_bufs = [t.view(torch.uint8) for t in recv_bufs]
        dist.gather(send, recv_bufs, dst=dst)
        raw = local_full.view(torch.uint8)
        for r, (rb, re) in enumerate(ranges):
            if r != dst and re > rb:
                raw[rb * item:re * item] = recv_bufs[r][:(re - rb) * item].to(local_full.device)
    else:
        dist.gather(send, None, dst=dst)
    return local_full


def gather_locations(loc, rank: int, world: int, dist, dst: int = 0, device=None):
    """csv / --exclude-pseudo across ranks: `loc` = (pos_begin, plus_off, plus, minus_off, minus) of this rank's shard
    (Index.locate(..., kmer_range=ranges[rank]); occurrence lists per slice position, variable length).
    Counts first, then payload: one all_gather of the four sizes, one gather of a single padded int64 buffer
    [plus_off | minus_off | plus | minus].  Returns the merged tuple on `dst` (shards are contiguous and ordered by rank),
    None elsewhere.  `device`: where the exchanged tensors live (a cuda device for backend nccl = RCCL; None = host)."""
    import numpy as np
    import torch
    pos_begin, po, pl, mo, mi = loc
    n_pos = len(po) - 1
    if world == 1:
        return loc
    sizes = torch.tensor([int(pos_begin), n_pos, len(pl), len(mi)], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [[int(v) for v in t.cpu().tolist()] for t in all_sizes]
    need = max(2 * (s[1] + 1) + s[2] + s[3] for s in all_sizes)
    buf = np.zeros(need, dtype=np.int64)
    k = 0
    for a in (po, mo, pl, mi):
        buf[k:k + len(a)] = np.asarray(a).view(np.int64) if len(a) else 0
        k += len(a)
    send = torch.from_numpy(buf).to(device) if device is not None else torch.from_numpy(buf)
    recv = [torch.empty(need, dtype=torch.int64, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    out_po, out_mo, out_pl, out_mi = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)], [], []
    base_p = base_m = 0
    begin = None
    expect = None
    for r, (pb, npos, npl, nmi) in enumerate(all_sizes):
        if npos == 0:
            continue
        b = recv[r].cpu().numpy().view(np.uint64)
        rpo, rmo = b[:npos + 1], b[npos + 1:2 * (npos + 1)]
        rpl, rmi = b[2 * (npos + 1):2 * (npos + 1) + npl], b[2 * (npos + 1) + npl:2 * (npos + 1) + npl + nmi]
        if begin is None:
            begin = pb
        elif pb != expect:
            raise ValueError(f"location shards are not contiguous: rank {r} begins at {pb}, expected {expect}")
        expect = pb + npos
        out_po.append(rpo[1:] + np.uint64(base_p)); out_mo.append(rmo[1:] + np.uint64(base_m))
        out_pl.append(rpl); out_mi.append(rmi)
        base_p += npl; base_m += nmi
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint64)
    return (0 if begin is None else begin), cat(out_po), cat(out_pl), cat(out_mo), cat(out_mi)
