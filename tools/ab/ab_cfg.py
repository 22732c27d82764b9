#!/usr/bin/env python3
"""A/B of builds of the library on ONE box: tools/ab/ab_cfg.py <lib.so or ''> K,E,reps[,kmer_begin,kmer_end] ...  (3.09 Gbp text; search-kernel ms)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import genmap_amd as g
import genmap_amd.capi as capi
from genmap_amd import synth
lib = sys.argv[1]
if lib:
    p = Path(lib).resolve()
    capi.lib_path = lambda profiling=False: p
codes, lens, desc = synth.workload("grch38", 1.0)
ix = g.Index.build(codes, lens, sampling=1)
out = torch.zeros(len(codes) + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for cfg in sys.argv[2:]:
    f = [int(x) for x in cfg.split(",")]
    K, E, reps = f[:3]
    rng = (f[3], f[4]) if len(f) >= 5 else None
    for _ in range(reps + 1):
        ix.map_device(out.data_ptr(), K, E, value_bits=8, kmer_range=rng, stream=st)
    ms = ix.kernel_times(reps)
    chk = int(out[:len(codes)].to(torch.int64).sum().item())
    print(f"{lib or 'current tree'}: K={K} E={E}{(' on %.2f G k-mers' % ((rng[1] - rng[0]) / 1e9)) if rng else ''}: search kernel min {min(ms):.2f} ms, mean {sum(ms) / len(ms):.2f} ms  checksum {chk}", flush=True)
ix.close()
