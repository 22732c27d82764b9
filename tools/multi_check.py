"""2-rank rehearsal on one GPU: shards computed by two processes and gathered == the unsharded result."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.distributed as dist
import genmap_amd as g
from genmap_amd import synth
from genmap_amd.distributed import gather_frequency, max_shard_len, shard_ranges
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
codes, lens, _ = synth.workload("chr1", 0.02)
ix = g.Index.build(codes, lens, sampling=1)
ok = True
for K, E in ((30, 0), (30, 1), (100, 1)):
    n = len(codes); nk = n - K + 1
    step = K - g.tuned_infix_length(K, E) + 1
    ranges = shard_ranges(nk, step, world); m = max_shard_len(ranges)
    out = torch.zeros(n + m, dtype=torch.uint8, device="cuda:0")
    ix.map_device(out.data_ptr(), K, E, value_bits=8, kmer_range=ranges[rank])
    torch.cuda.synchronize()
    gather_frequency(out, ranges, rank, world, dist, stage_on_host=True)
    if rank == 0:
        full = ix.map(K, E, value_bits=8)
        same = np.array_equal(out[:n].cpu().numpy(), full)
        print(f"K={K} E={E}: gathered == unsharded: {same}")
        ok &= same
dist.barrier()
if rank == 0:
    print("MULTI_OK" if ok else "MULTI_FAIL")
dist.destroy_process_group()
