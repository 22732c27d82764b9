#!/bin/bash
# end-to-end `genmap` at scale: FASTA -> index directory -> map (-r -fs and -bg), timed, result checked against the library path
export TMPDIR=/tmp
D=/tmp/gmcli; rm -rf $D; mkdir -p $D/out
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from genmap_amd import synth
codes, lens, desc = synth.workload("chr1", 0.2)
lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
seq = lut[codes]
with open("/tmp/gmcli/genome.fa", "wb") as f:
    f.write(b">chr1 synthetic\n")
    for i in range(0, len(seq), 60 * 100000):
        chunk = seq[i:i + 60 * 100000]
        n = len(chunk) // 60 * 60
        f.write(b"\n".join(bytes(r) for r in chunk[:n].reshape(-1, 60)) + b"\n")
        if n < len(chunk): f.write(bytes(chunk[n:]) + b"\n")
print(desc)
PY
ls -la $D/genome.fa
TIMEFORMAT="index: %R s"; time genmap_amd/bin/genmap index -F $D/genome.fa -I $D/idx -v 2>&1 | tail -4
du -sh $D/idx
TIMEFORMAT="map e0 raw: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 0 -r -fs -v 2>&1 | tail -4
TIMEFORMAT="map e0 bedgraph (GPU runs): %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 0 -bg -fs -v 2>&1 | tail -3
TIMEFORMAT="map e2 raw: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 2 -r -fs -v 2>&1 | tail -3
ls -la $D/out | head; head -3 $D/out/genome.genmap.bedgraph
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import genmap_amd as g
from genmap_amd import synth
codes, lens, desc = synth.workload("chr1", 0.2)
ix = g.Index.build(codes, lens, sampling=1)
ref = ix.map(30, 2, value_bits=8)
got = np.fromfile("/tmp/gmcli/out/genome.genmap.freq8", dtype=np.uint8)
print("CLI freq8 (E=2) == library:", np.array_equal(ref, got), len(got))
PY
