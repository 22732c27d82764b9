#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: tools/ab/ab_e0.py <lib.so or ''> -- K=30 e=0 and K=100 e=1 kernel times on the 3.09 Gbp text."""
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
codes, lens, desc = synth.workload("grch38", 1.0)
ix = g.Index.build(codes, lens, sampling=1)
out = torch.zeros(len(codes) + 16, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for K, E, reps, rng in ((30, 0, 12, None), (100, 1, 4, None), (30, 1, 3, (1200000000, 1800000000)), (30, 2, 2, (1500000000, 1590000000))):
    for _ in range(reps + 1):
        ix.map_device(out.data_ptr(), K, E, value_bits=8, kmer_range=rng, stream=st)
    ms = ix.kernel_times(reps)
    print(f"{sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else 'current tree'}: K={K} E={E}{(' on %.2f G k-mers' % ((rng[1] - rng[0]) / 1e9)) if rng else ''}: search kernel min {min(ms):.2f} ms, mean {sum(ms) / len(ms):.2f} ms", flush=True)
ix.close()
