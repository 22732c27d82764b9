#!/usr/bin/env python3
"""Build one index, then time gm_map_device under several settings of the library's scheduling knobs (gm_index_set_tuning).
Usage: sweep_tuning.py --workload grch38 --cfg 30,0,1.0 30,2,0.06 -- "" "skip_dup=1" "skip_dup=1,fetch_batch=16" ...
A cfg is K,E,frac (frac = share of the k-mers: a contiguous range from the middle of the text).  INFIX=n and STEP=n (k-mers per block, infix = K - n + 1) are pseudo-knobs."""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import genmap_amd as g
from genmap_amd import synth

DEFAULTS = dict(verify_t=-1, lds_stack=-1, blocks_per_cu=4, qtable=-1, sat_min_w=256, fetch_batch=-1, probation=-1, verify_cost=3, no_store=0, no_saturate=0, skip_dup=-1, coop=-1, use_ctx=1, steal=-1, part_bias=0, oss_weights=-1, child_tables=-1, jump=-1, self_hit=1, jump_filter=1, range_add=1, verify_t_ext=-1, jump_groups=-1, iter_cap=-1, stall_cap=-1, fast_verify=-1, no_wrap=-1, jump_layouts=-1, lds_pad=0, pat_batch=-1, expand=-1, expand_mb=-1, expand_chunk=-1, sat_draw_w=-1, expand_occ=-1, expand_overlap=-1, expand_two_pass=-1, expand_share=-1, win2=-1)
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="chr1"); ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--cfg", nargs="+", default=["30,2,1.0"])
ap.add_argument("--reps", type=int, default=2); ap.add_argument("--block-bytes", type=int, default=0); ap.add_argument("--sampling", type=int, default=1)
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
if a.workload == "bacteria5":   # config C5's multi-genome text, as ONE frequency call over all five files
    import numpy as np
    recs = [c for _, rs in synth.bacteria5(a.scale) for _, c in rs]
    codes, lens, desc = np.concatenate(recs), [len(c) for c in recs], f"S5 five-bacteria-like {sum(len(c) for c in recs)} bp in {len(recs)} sequences"
else:
    codes, lens, desc = synth.workload(a.workload, a.scale)
t0 = time.time(); ix = g.Index.build(codes, lens, sampling=a.sampling, block_bytes=a.block_bytes); print(f"{desc}: index in {time.time()-t0:.1f} s", flush=True)
n = len(codes)
out = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for cfg in a.cfg:
    K, E, frac = cfg.split(","); K, E, frac = int(K), int(E), float(frac)
    nk = n - K + 1
    base = None
    for st in a.settings:
        knobs = dict(DEFAULTS); infix = 0
        for kv in filter(None, st.split(",")):
            k, v = kv.split("=")
            if k == "INFIX": infix = int(v)
            elif k == "STEP": infix = K - int(v) + 1   # k-mers per block
            else: knobs[k] = int(v)
        ix.set_tuning(**knobs)
        step = K - (infix or g.tuned_infix_length(K, E)) + 1
        span = int(nk * frac) // step * step
        kb = ((nk - span) // 2) // step * step
        rng = None if frac >= 1.0 else (kb, kb + span)
        out.zero_()
        try:
            for r in range(a.reps + 1):
                ix.map_device(out.data_ptr(), K, E, infix=infix, value_bits=8, kmer_range=rng, stream=stream)
        except g.GenmapError as ex:   # a setting that does not apply to this (K, E)
            print(f"K={K} E={E} frac={frac:<5} {st:45s} skipped: {ex}", flush=True)
            continue
        ms = ix.kernel_times(a.reps)
        chk = int(out[:n].to(torch.int64).sum().item())
        if base is None: base = chk
        best = min(ms)
        det = ix.last_stats()["detail"]
        corr = det.get("correction_us", 0) / 1e3; tq = det.get("table_q", 0)
        print(f"K={K} E={E} frac={frac:<5} {st or '(default)':45s} {best:10.2f} ms (correction pass {corr:8.2f})  {(span if rng else nk)/best*1e3:10.4g} k-mers/s  checksum {'ok' if chk == base else 'DIFFERS'}  q={tq & 255} J={tq >> 8}", flush=True)
ix.close()
