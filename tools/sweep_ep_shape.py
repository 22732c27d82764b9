#!/usr/bin/env python3
"""Config C5's --exclude-pseudo pass (five FASTA files, K=24 e=1) under several block shapes (k-mers per block) and knob settings:
product library, search-kernel ms summed over the five files.  tools/sweep_ep_shape.py [K E] -- "" "verify_t=0" ..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import genmap_amd as g
from genmap_amd import synth

args = sys.argv[1:]
shapes = (0, 1, 2, 3, 4, 6, 8)
if "--shapes" in args:   # e.g. --shapes 0 : the library's block shape only (knob sweeps)
    k = args.index("--shapes"); shapes = tuple(int(x) for x in args[k + 1].split(",")); args = args[:k] + args[k + 2:]
sets = [""]
if "--" in args:
    k = args.index("--"); sets = args[k + 1:] or [""]; args = args[:k]
K, E = (int(args[0]), int(args[1])) if len(args) >= 2 else (24, 1)
files = synth.bacteria5(1.0)
recs = [c for _, rs in files for _, c in rs]
codes, lens = np.concatenate(recs), [len(c) for c in recs]
fid = np.concatenate([[i] * len(rs) for i, (_, rs) in enumerate(files)]).astype(np.uint32)
ix = g.Index.build(codes, lens, sampling=1)
out = torch.zeros(len(codes) + 16, dtype=torch.uint16, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
ref = None
for setting in sets:
    knobs = dict(verify_t=-1, steal=-1, fetch_batch=-1, probation=-1, lds_stack=-1, blocks_per_cu=4, verify_t_ext=-1, sat_min_w=256, skip_dup=-1, qtable=-1)
    for kv in filter(None, setting.split(",")):
        a, b = kv.split("="); knobs[a] = int(b)
    ix.set_tuning(**knobs)
    for n in shapes:
        infix = 0 if n == 0 else K - n + 1
        tot = 0.0; res = []
        for rep in range(2):
            tot = 0.0; first = 0; res = []
            for f, (_, rs) in enumerate(files):
                ix.map_device(out.data_ptr(), K, E, first_seq=first, n_seq=len(rs), infix=infix, value_bits=16, exclude_pseudo=True, seq_file_id=fid, stream=st)
                tot += ix.kernel_times(1)[-1]
                tl = sum(len(c) for c in rs)
                res.append(out[:tl].cpu().numpy().copy())
                first += len(rs)
        chk = int(sum(int(r.astype(np.int64).sum()) for r in res))
        if ref is None: ref = chk
        print(f"K={K} E={E} {setting or '(default)':24s} k-mers per block {n if n else 'library':>7}: search kernels of the five files {tot:8.2f} ms  checksum {'ok' if chk == ref else 'DIFFERS'}", flush=True)
ix.close()
