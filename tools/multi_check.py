"""2- or 3-rank rehearsal on ONE GPU (torchrun, backend gloo): the multi-GPU data paths with every rank on cuda:0.
  * interleaved chunks (ShardPlan) computed by the ranks, gathered with the collective  == the unsharded result
  * the same through PeerGather: the root's vector shared over HIP IPC, chunks pushed with device-to-device copies on a copy
    stream while the next launch computes (between GPUs these copies go over xGMI; here they stay on the device)
  * config C5's shape: --exclude-pseudo frequencies and csv location lists as contiguous shares"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.distributed as dist
import genmap_amd as g
from genmap_amd import synth
from genmap_amd.distributed import PeerGather, ShardPlan, gather_chunks, gather_frequency, max_shard_len, shard_ranges
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
codes, lens, _ = synth.workload("chr1", 0.02)
ix = g.Index.build(codes, lens, sampling=1)
ok = True
stream = torch.cuda.current_stream().cuda_stream
for K, E in ((30, 0), (30, 1), (100, 1)):
    n = len(codes); nk = n - K + 1
    plan = ShardPlan(nk, K - g.tuned_infix_length(K, E) + 1, world, chunks_per_rank=9)
    full = ix.map(K, E, value_bits=8) if rank == 0 else None
    # (a) collective gather of the chunks
    out = torch.zeros(plan.padded_len(n), dtype=torch.uint8, device="cuda:0")
    ix.map_device(out.data_ptr(), K, E, value_bits=8, chunks=plan.chunk_arg(rank), stream=stream)
    torch.cuda.synchronize()
    gather_chunks(out, plan, rank, dist, stage_on_host=True)
    if rank == 0:
        same = np.array_equal(out[:n].cpu().numpy(), full)
        print(f"K={K} E={E}: chunks gathered with the collective == unsharded: {same}")
        ok &= same
    # (b) peer copies into the root's vector, launch by launch
    pg = PeerGather(plan, n, 1, rank, 0, dist, launches=3, piece_bytes=3 * plan.chunk_len * world)   # several pieces even at this size
    if not pg.ok:
        print("PeerGather: IPC not available here"); ok = False
    else:
        for sub in plan.sub_ranges(pg.launches):
            ix.map_device(pg.local_ptr, K, E, value_bits=8, kmer_range=sub, chunks=plan.chunk_arg(rank), stream=stream)
            ev = torch.cuda.Event(); ev.record()
            pg.push(sub, ev)
        torch.cuda.synchronize(); pg.finish(); dist.barrier()
        if rank == 0:
            got = torch.empty(pg.nbytes, dtype=torch.uint8, device="cuda:0")
            pg.assemble(got.data_ptr())
            torch.cuda.synchronize()
            got = got[:n]
            same = np.array_equal(got.cpu().numpy(), full)
            print(f"K={K} E={E}: chunks pushed over IPC peer copies == unsharded: {same}")
            ok &= same
    pg.close()
# config C5's shape: several FASTA files, --exclude-pseudo frequencies and csv location lists, sharded and gathered
from genmap_amd.distributed import gather_locations
files5 = synth.bacteria5(0.05)
codes5 = np.concatenate([c for _, recs in files5 for _, c in recs]); lens5 = [len(c) for _, recs in files5 for _, c in recs]
fid = np.array([f for f, (_, recs) in enumerate(files5) for _ in recs], dtype=np.uint32)
ix5 = g.Index.build(codes5, lens5, sampling=1)
K, E = 24, 1
n5 = len(codes5); nk5 = n5 - K + 1
step5 = K - g.tuned_infix_length(K, E) + 1
ranges5 = shard_ranges(nk5, step5, world); m5 = max_shard_len(ranges5)
out5 = torch.zeros(n5 + m5, dtype=torch.uint16, device="cuda:0")
ix5.map_device(out5.data_ptr(), K, E, value_bits=16, exclude_pseudo=True, seq_file_id=fid, kmer_range=ranges5[rank])
torch.cuda.synchronize()
gather_frequency(out5, ranges5, rank, world, dist, stage_on_host=True)
loc = ix5.locate(K, E, kmer_range=ranges5[rank])
merged = gather_locations(loc, rank, world, dist)
if rank == 0:
    same = np.array_equal(out5[:n5].cpu().numpy(), ix5.map(K, E, value_bits=16, exclude_pseudo=True, seq_file_id=fid))
    print(f"-ep K={K} E={E}: gathered == unsharded: {same}"); ok &= same
    full = ix5.locate(K, E)
    same = merged[0] == full[0] and all(np.array_equal(a, b) for a, b in zip(merged[1:], full[1:]))
    print(f"csv locations K={K} E={E}: gathered == unsharded: {same} ({len(full[2])} + {len(full[4])} occurrences)"); ok &= same
ix5.close()
dist.barrier()
if rank == 0:
    print("MULTI_OK" if ok else "MULTI_FAIL")
dist.destroy_process_group()
