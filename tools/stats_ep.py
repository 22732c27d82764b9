#!/usr/bin/env python3
"""Device-side step statistics of --exclude-pseudo calls on config C5's text (instrumented twin), per block shape."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import genmap_amd as g
from genmap_amd import synth

files = synth.bacteria5(1.0)
recs = [c for _, rs in files for _, c in rs]
codes, lens = np.concatenate(recs), [len(c) for c in recs]
fid = np.concatenate([[i] * len(rs) for i, (_, rs) in enumerate(files)]).astype(np.uint32)
ix = g.Index.build(codes, lens, sampling=1, profiling=True)
K, E = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1
nfirst = len(files[0][1])
for infix in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["24", "20", "17"])]:
    for ep in (True, False):
        ix.map(K, E, first_seq=0, n_seq=nfirst, infix=infix, value_bits=16, exclude_pseudo=ep, seq_file_id=fid if ep else None)
        st = ix.last_stats(); d = st["detail"]; nk = st["kmers"]
        print(json.dumps({"K": K, "E": E, "infix": infix, "n": K - infix + 1, "ep": ep, "kmers": nk, "search_ms_twin": round(st["search_ms"], 2), "steps_per_kmer": round(st["node_steps"] / nk, 2), "lines_per_kmer": round(st["rank_lines"] / nk, 2),
                          "verify_items_per_kmer": round(d["verify_items"] / nk, 3), "chunks_per_item": round(d["verify_chunks"] / max(1, d["verify_items"]), 2), "pushes_per_step": round(d["pushes"] / max(1, st["node_steps"]), 3),
                          "wave_iterations": d["wave_iterations"], "lanes": round(d["active_lane_sum"] / max(1, d["wave_iterations"]), 1), "oss_frac": round(d["steps_oss"] / max(1, st["node_steps"]), 3),
                          "passes": {k[2:]: round(d[k] / max(1, d["wave_iterations"]), 3) for k in d if k.startswith("w_")}}), flush=True)
ix.close()
