"""One-process-per-GPU sharding of a computeMappability call and the gather of its result.

Every rank holds a full index replica (a 3.1 Gbp index is a few GB of the 288) and computes disjoint sets of whole k-mer
blocks; positions are independent given the read-only index (the only coupling in the reference, copying a value to
duplicate k-mers at src/algo.hpp:389-396, is an optimisation the GPU build drops).

Load balance: the reference deals >= 50 dynamic chunks per worker because "repeats are slower than unique regions"
(src/algo.hpp:422-434).  ShardPlan does the static equivalent: the k-mer blocks are cut into fixed chunks -- CHUNKS_PER_RANK
per rank -- dealt round-robin, so an 18-Mbp N desert or a repeat family is spread over every rank
(gm_map_params.chunk_blocks / chunk_index / chunk_stride).  The only collective is ONE gather of the ranks' chunks of the
8/16-bit frequency vector to the root over RCCL/xGMI -- no reduction, positions are disjoint.  Works with any
torch.distributed backend (nccl = RCCL on ROCm; gloo in the CPU tests).  csv / --exclude-pseudo location lists are
variable-length and travel as contiguous ranges (shard_ranges + gather_locations)."""
from typing import List, Tuple

CHUNKS_PER_RANK = 64


class ShardPlan:
    """Interleaved chunks of whole k-mer blocks for `world` ranks over the k-mer positions [0, num_kmers)."""

    def __init__(self, num_kmers: int, step_size: int, world: int, chunks_per_rank: int = CHUNKS_PER_RANK):
        self.num_kmers, self.step_size, self.world = int(num_kmers), int(step_size), int(world)
        self.nblocks = (self.num_kmers + step_size - 1) // step_size
        want = max(1, world * chunks_per_rank)
        self.chunk_blocks = max(1, -(-self.nblocks // want))            # ceil: at most `want` chunks
        self.chunk_len = self.chunk_blocks * step_size                   # positions per chunk
        self.nchunks = -(-self.nblocks // self.chunk_blocks) if self.nblocks else 0
        self.rows = -(-max(self.nchunks, 1) // world) * world            # chunk slots, a multiple of world

    def chunk_arg(self, rank: int):
        """(chunk_blocks, chunk_index, chunk_stride) for Index.map_device; None on a single rank"""
        return None if self.world == 1 else (self.chunk_blocks, rank, self.world)

    def padded_len(self, text_len: int) -> int:
        """elements a rank's result buffer needs so that it can be viewed as [rows, chunk_len]"""
        return max(int(text_len), self.rows * self.chunk_len) + 16

    def positions_of(self, rank: int):
        """[(begin, end)] position ranges owned by `rank` (tests, diagnostics)"""
        out = []
        for c in range(rank, self.nchunks, self.world):
            b = c * self.chunk_len
            out.append((b, min(b + self.chunk_len, self.num_kmers)))
        return out

    def sub_ranges(self, launches: int):
        """the k-mer positions cut into `launches` row-aligned ranges (a row = one chunk of every rank): the pieces of a pass
        whose results travel while the next piece is computed"""
        row_len = self.chunk_len * self.world
        rows = self.rows // self.world if self.world else 0
        rows = max(rows, 1)
        out = []
        for s in range(launches):
            b, e = rows * s // launches * row_len, rows * (s + 1) // launches * row_len
            b, e = min(b, self.num_kmers), min(e, self.num_kmers)
            if e > b:
                out.append((b, e))
        return out or [(0, self.num_kmers)]

    def describe(self) -> str:
        if self.world == 1:
            return "one GPU, whole text"
        return (f"{self.nchunks} chunks of {self.chunk_blocks} k-mer blocks ({self.chunk_len} positions) dealt round-robin to {self.world} ranks, "
                f"index replicated, one gather of the 8-bit chunks to rank 0")


class PiecePlan:
    """How the gathered vector is cut into separately allocated PIECES of whole chunk rows (a row = one chunk slot of every rank),
    and which strided copies carry a rank's chunks of a row-aligned k-mer range into them.  Pure arithmetic (CPU-testable)."""

    def __init__(self, plan: ShardPlan, text_len: int, item_bytes: int, piece_bytes: int):
        self.plan, self.item = plan, item_bytes
        self.nbytes = plan.padded_len(text_len) * item_bytes
        self.row_bytes = plan.chunk_len * plan.world * item_bytes
        nrows = max(1, plan.rows // plan.world)
        self.rows_per_piece = max(1, piece_bytes // max(1, self.row_bytes))
        self.piece_off = [r * self.row_bytes for r in range(0, nrows, self.rows_per_piece)]
        # the last piece runs to the end of the padded vector (the K - 1 zeros behind the last k-mer may lie past the last row)
        self.piece_len = [(self.piece_off[i + 1] if i + 1 < len(self.piece_off) else self.nbytes) - o for i, o in enumerate(self.piece_off)]

    def locate(self, byte_off: int):
        """(piece number, offset inside it) of a byte offset of the vector"""
        q = min(byte_off // (self.rows_per_piece * self.row_bytes), len(self.piece_off) - 1)
        return q, byte_off - self.piece_off[q]

    def copies(self, rank: int, sub_range):
        """[(piece, first byte inside the piece, first byte of the vector, pitch, bytes per chunk, chunks)]: the chunks of `rank`
        inside sub_range, as runs of rows that lie in one piece"""
        plan, item = self.plan, self.item
        row_len = plan.chunk_len * plan.world
        r0, r1 = sub_range[0] // row_len, -(-sub_range[1] // row_len)
        rows = [c for c in range(r0, r1) if c * plan.world + rank < plan.nchunks]
        out, k = [], 0
        while k < len(rows):
            q = rows[k] // self.rows_per_piece
            e = k
            while e < len(rows) and rows[e] // self.rows_per_piece == q:
                e += 1
            first = (rows[k] * plan.world + rank) * plan.chunk_len * item
            out.append((q, first - self.piece_off[q], first, row_len * item, plan.chunk_len * item, e - k))
            k = e
        return out


class PeerGather:
    """The root's result vector shared through HIP IPC handles; every rank pushes its finished chunks into it with
    device-to-device copies (xGMI, DMA engines) on a copy stream of its own while its search kernel works on the next chunks.

      pg = PeerGather(plan, text_len, item_bytes, rank, local_device, dist)      # collective; pg.ok False -> use gather_chunks
      for sub in plan.sub_ranges(pg.launches):                                    # every step
          ix.map_device(pg.local_ptr, ..., kmer_range=sub, chunks=plan.chunk_arg(rank), stream=compute)
          pg.push(sub, compute_stream_event)                                      # async copies of the chunks just computed
      pg.finish()                                                                 # wait for this rank's copies
      (root, after a barrier) pg.assemble(dst_ptr)                                # the gathered vector as one contiguous array

    The shared vector is NOT one allocation: opening an IPC handle of a 3.09 GB allocation never returned on MI355X / ROCm 7.2
    (profiles/r03/rehearsal_2rank_3p09gbp_diag.txt; 0.77 GB works), so the root allocates PIECES of whole chunk rows, at most
    `piece_bytes` each, and exports one handle per piece.  Every rank -- the root too -- computes into a contiguous vector of
    its own (local_ptr) and copies its chunks into the pieces.  An open that does not return within `ipc_timeout` seconds, or a
    probe pattern that does not arrive, makes every rank fall back to the RCCL gather (ok == False on all ranks)."""

    PIECE_BYTES = 512 << 20

    def __init__(self, plan: ShardPlan, text_len: int, item_bytes: int, rank: int, device: int, dist, launches: int = 4, mark=None,
                 piece_bytes: int = 0, ipc_timeout: float = 60.0):
        import threading
        import torch
        from . import capi
        mark = mark or (lambda *a: None)
        self.plan, self.rank, self.device, self.item, self.dist = plan, rank, device, item_bytes, dist
        self.launches = max(1, min(launches, plan.rows // plan.world if plan.world else 1))
        self.nbytes = plan.padded_len(text_len) * item_bytes
        self.ok = False
        self.local_ptr = capi.device_alloc(device, self.nbytes)      # this rank's own contiguous vector: what its kernel writes
        self.copy_stream = torch.cuda.Stream(device=device)
        self.pieces = PiecePlan(plan, text_len, item_bytes, piece_bytes or self.PIECE_BYTES)
        self.piece_off, self.piece_len = self.pieces.piece_off, self.pieces.piece_len
        self.piece_ptr: List[int] = []
        self._opened: List[int] = []
        mark(f"PeerGather: {self.nbytes} bytes in {len(self.piece_off)} pieces of at most {max(self.piece_len)} bytes")
        good = 1
        handles = [None]
        if rank == 0:
            try:   # only the root's allocation and export may fail here; the broadcast below ALWAYS runs (a root that skipped it left
                   # the other ranks inside it while it went on to the all_reduce: mismatched collectives)
                for n in self.piece_len:   # (appended one by one: a failure half way must not lose the pieces already allocated)
                    self.piece_ptr.append(capi.device_alloc(device, n))
                handles = [[capi.ipc_export(device, q) for q in self.piece_ptr]]
            except Exception as e:   # noqa: BLE001
                mark(f"PeerGather: the root cannot allocate / export its pieces: {e}")
                for q in self.piece_ptr:
                    try:
                        capi.device_free(device, q)
                    except Exception:   # noqa: BLE001
                        pass
                self.piece_ptr = []
                handles = [None]
                good = 0
        dist.broadcast_object_list(handles, src=0)
        mark("PeerGather: handles exchanged")
        if handles[0] is None:      # the root failed: every rank falls back
            good = 0
        try:
            if rank != 0 and good:
                box = {}

                def _open():
                    try:
                        box["ptrs"] = [capi.ipc_open(device, h) for h in handles[0]]
                    except Exception as e:   # noqa: BLE001 -- any failure means "no peer copies here"
                        box["err"] = e
                th = threading.Thread(target=_open, daemon=True)
                th.start()
                th.join(ipc_timeout)
                if th.is_alive() or "ptrs" not in box:
                    mark(f"PeerGather: IPC open {'did not return in %.0f s' % ipc_timeout if th.is_alive() else 'failed: %s' % box.get('err')}")
                    good = 0
                else:
                    self.piece_ptr = self._opened = box["ptrs"]
                    mark("PeerGather: IPC handles opened")
        except Exception as e:   # noqa: BLE001
            mark(f"PeerGather: IPC setup failed: {e}")
            good = 0
        # pattern exchange: rank r writes r + 1 into the first bytes of its first chunk and of its last one
        probes = []
        if good and plan.nchunks > rank:
            mine = list(range(rank, plan.nchunks, plan.world))
            probes = sorted({mine[0], mine[-1]})
        if good and rank != 0 and probes:
            t = torch.full((min(64, plan.chunk_len * item_bytes),), rank + 1, dtype=torch.uint8, device=f"cuda:{device}")
            try:
                for c in probes:
                    q, off = self._locate(c * plan.chunk_len * item_bytes)
                    capi.push_pieces(device, self.piece_ptr[q] + off, t.data_ptr(), 0, 0, t.numel(), 1, 0, self.copy_stream.cuda_stream)
                self.copy_stream.synchronize()
                mark("PeerGather: probe patterns pushed")
            except Exception as e:   # noqa: BLE001
                mark(f"PeerGather: probe push failed: {e}")
                good = 0
        flag = torch.tensor([good], dtype=torch.int32)
        cpu_ok = dist.get_backend() != "nccl"
        flag = flag if cpu_ok else flag.to(f"cuda:{device}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dist.barrier()
        if int(flag.item()) == 1 and rank == 0:
            for r in range(1, plan.world):
                if plan.nchunks <= r:
                    continue
                mine = list(range(r, plan.nchunks, plan.world))
                for c in sorted({mine[0], mine[-1]}):
                    q, off = self._locate(c * plan.chunk_len * item_bytes)
                    probe = torch.empty(1, dtype=torch.uint8, device=f"cuda:{device}")
                    capi.push_pieces(device, probe.data_ptr(), self.piece_ptr[q] + off, 0, 0, 1, 1, 0, None)
                    torch.cuda.synchronize()
                    if int(probe.item()) != r + 1:
                        flag[0] = 0
                    capi.push_pieces(device, self.piece_ptr[q] + off, self.local_ptr, 0, 0, min(64, plan.chunk_len * item_bytes), 1, 0, None)   # zeros again
            torch.cuda.synchronize()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self.ok = int(flag.item()) == 1
        mark(f"PeerGather: probes checked, ok={self.ok}")

    def _locate(self, byte_off: int):
        return self.pieces.locate(byte_off)

    def push(self, sub_range, after_event):
        """copy this rank's chunks inside sub_range (row-aligned k-mer range) into the root's pieces, after `after_event`"""
        from . import capi
        todo = self.pieces.copies(self.rank, sub_range)
        if not todo:
            return
        self.copy_stream.wait_event(after_event)
        for q, in_piece, in_vector, pitch, nbytes, count in todo:   # dst and src advance by the same offsets
            capi.push_pieces(self.device, self.piece_ptr[q], self.local_ptr + (in_vector - in_piece), in_piece, pitch, nbytes, count, 0,
                             self.copy_stream.cuda_stream)

    def finish(self):
        self.copy_stream.synchronize()

    def assemble(self, dst_ptr: int, stream=None):
        """root: the pieces as ONE contiguous vector of nbytes at dst_ptr (device-to-device, after every rank has finished)"""
        from . import capi
        assert self.rank == 0
        for q, off in enumerate(self.piece_off):
            capi.push_pieces(self.device, dst_ptr + off, self.piece_ptr[q], 0, 0, self.piece_len[q], 1, 0, stream)

    def close(self):
        from . import capi
        try:
            for q in self._opened:
                capi.ipc_close(self.device, q)
        finally:
            self.dist.barrier()
            if self.rank == 0:
                for q in self.piece_ptr:
                    capi.device_free(self.device, q)
            capi.device_free(self.device, self.local_ptr)


def gather_chunks(local_full, plan: ShardPlan, rank: int, dist, dst: int = 0, stage_on_host: bool = False):
    """local_full: this rank's frequency vector (1-D tensor of plan.padded_len(text_len) elements, own chunks filled).
    After the call the dst rank's local_full holds every rank's chunks.  The rank's chunks are a strided view
    [rank::world] of the [rows, chunk_len] matrix: packed with one device copy, gathered, unpacked the same way."""
    world = plan.world
    if world == 1:
        return local_full
    import torch
    item = local_full.element_size()
    mat = local_full[:plan.rows * plan.chunk_len].view(torch.uint8).view(plan.rows, plan.chunk_len * item)   # raw bytes: collectives do not take 16-bit unsigned
    send = mat[rank::world].contiguous()
    if stage_on_host:   # backends without device-memory collectives (gloo rehearsal): same data path through host buffers
        send = send.cpu()
    if rank == dst:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, recv, dst=dst)
        for r in range(world):
            if r != dst:
                mat[r::world] = recv[r].to(local_full.device)
    else:
        dist.gather(send, None, dst=dst)
    return local_full


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
