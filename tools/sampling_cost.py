"""What -S costs on the GPU: config C5's shape (5 related genomes, K=24 E=1) with the full suffix array and with sampled ones.
--exclude-pseudo frequencies (locate of every hit of a multi-hit k-mer) and plain frequencies (no locate: the difference is the
text verification of narrow nodes, which needs the full array)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import genmap_amd as g
from genmap_amd import synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
files = synth.bacteria5(scale)
codes = np.concatenate([c for _, recs in files for _, c in recs]); lens = [len(c) for _, recs in files for _, c in recs]
fid = np.array([f for f, (_, recs) in enumerate(files) for _ in recs], dtype=np.uint32)
K, E = 24, 1
ref = None
for s in (1, 2, 10, 64):
    t0 = time.time(); ix = g.Index.build(codes, lens, sampling=s); tb = time.time() - t0
    info = ix.info()
    row = [f"sampling {s:2d}: build {tb:5.2f} s, index {info['device_bytes'] / 2**20:8.1f} MiB"]
    for ep in (False, True):
        kw = dict(value_bits=16, exclude_pseudo=ep, seq_file_id=fid if ep else None)
        out = ix.map(K, E, **kw)
        ms = []
        for _ in range(3):
            t0 = time.time(); ix.map(K, E, **kw); ms.append((time.time() - t0) * 1e3)
        row.append(f"{'-ep' if ep else 'freq'} gm_map {min(ms):8.2f} ms")
        if s == 1:
            ref = ref or {}; ref[ep] = out
        else:
            assert np.array_equal(out, ref[ep]), (s, ep)
    t0 = time.time(); loc = ix.locate(K, E, kmer_range=(0, 200000)); tl = time.time() - t0
    row.append(f"csv locate of 200k positions {tl * 1e3:7.1f} ms ({len(loc[2]) + len(loc[4])} occurrences)")
    print(", ".join(row), flush=True)
    ix.close()
print("results identical for every sampling rate")
