#!/usr/bin/env python3
"""A/B of builds of the library on ONE box, K=30 e=0 only (the kernel that sits at its wall): tools/ab/ab_e0_only.py <lib.so or ''> [reps]."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import genmap_amd as g
import genmap_amd.capi as capi
from genmap_amd import synth
if len(sys.argv) > 1 and sys.argv[1]:
    p = Path(sys.argv[1]).resolve()
    capi.lib_path = lambda profiling=False: p
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
codes, lens, desc = synth.workload("grch38", 1.0)
ix = g.Index.build(codes, lens, sampling=1)
out = torch.zeros(len(codes) + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps + 1):
    ix.map_device(out.data_ptr(), 30, 0, value_bits=8, stream=st)
ms = ix.kernel_times(reps)
chk = int(out[:len(codes)].to(torch.int64).sum().item())
print(f"{sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else 'current tree'}: K=30 E=0: search kernel min {min(ms):.2f} ms, mean {sum(ms) / len(ms):.2f} ms  checksum {chk}", flush=True)
ix.close()
