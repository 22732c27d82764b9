"""RCCL in the same process as libgenmap_amd.so, world_size 1: the collectives bench.py's N>1 path issues (gather of the
uint8 shard on device tensors, all_reduce MAX of the timing, barrier) run on the stream the search kernel was launched on
and leave the frequency vector intact.  (Two ranks cannot share one device under RCCL; the 2-rank data path is rehearsed
with gloo in tools/multi_check.py and tests/test_distributed_cpu.py.)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch, torch.distributed as dist
import genmap_amd as g
from genmap_amd import synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
codes, lens, _ = synth.workload("chr1", 0.01)
ix = g.Index.build(codes, lens, sampling=1)
n = len(codes)
out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ok = True
for K, E in ((30, 0), (30, 1)):
    ix.map_device(out.data_ptr(), K, E, value_bits=8, stream=stream)
    recv = [torch.empty(n, dtype=torch.uint8, device=dev)]
    dist.gather(out[:n], recv, dst=0)            # same call gather_frequency makes on the root
    t = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    ref = ix.map(K, E, value_bits=8)
    ok &= bool(np.array_equal(recv[0].cpu().numpy(), ref)) and float(t.item()) == 1.5
    print(f"K={K} E={E}: gathered over RCCL == host result: {ok}")
ix.close()
dist.destroy_process_group()
print("RCCL_OK" if ok else "RCCL_FAIL")
sys.exit(0 if ok else 1)
