#!/usr/bin/env python3
"""Build one index, then time gm_map_device under several settings of the library's environment knobs
(read with getenv at every call).  Usage: sweep_env.py --workload grch38 --K 30 --E 2 --frac 0.1 "GM_PROBATION=0" "GM_PROBATION=2,GM_SAT_MINW=1" ..."""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import genmap_amd as g
from genmap_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="chr1"); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--K", type=int, default=30); ap.add_argument("--E", type=int, default=2)
ap.add_argument("--frac", type=float, default=1.0, help="fraction of the k-mers (a contiguous range from the middle of the text)")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
codes, lens, desc = synth.workload(a.workload, a.scale)
t0 = time.time(); ix = g.Index.build(codes, lens, sampling=1); print(f"{desc}: index in {time.time()-t0:.1f} s", flush=True)
n = len(codes); nk = n - a.K + 1
step = a.K - g.tuned_infix_length(a.K, a.E) + 1
span = int(nk * a.frac) // step * step
kb = ((nk - span) // 2) // step * step
rng = None if a.frac >= 1.0 else (kb, kb + span)
out = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
base = None
for st in a.settings:
    keys = []; infix = 0
    for kv in filter(None, st.split(",")):
        k, v = kv.split("=")
        if k == "INFIX": infix = int(v); continue     # pseudo-knob: common-infix length of the call
        os.environ[k] = v; keys.append(k)
    ms = []
    out.zero_()
    for r in range(a.reps + 1):
        ix.map_device(out.data_ptr(), a.K, a.E, infix=infix, value_bits=8, kmer_range=rng, stream=stream)
        torch.cuda.synchronize()
        ms.append(ix.last_stats()["search_ms"])
    chk = int(out[:n].to(torch.int64).sum().item())
    if base is None: base = chk
    best = min(ms[1:])
    print(f"{st or '(default)':50s} {best:10.2f} ms  {(span if rng else nk)/best/1e3:10.4g} k-mers/s  checksum {'ok' if chk == base else 'DIFFERS'}", flush=True)
    for k in keys: os.environ.pop(k, None)
ix.close()
