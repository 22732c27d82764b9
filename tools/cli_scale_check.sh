#!/bin/bash
# end-to-end `genmap` at scale: FASTA -> index directory -> map (-r -fs, -bg, e=2), load / compute / write timed separately (-v),
# result checked against the library path.   tools/cli_scale_check.sh [workload scale devices]   e.g.  grch38 1.0 0   |   chr1 0.2 0,0
WL=${1:-chr1}; SC=${2:-0.2}; DEV=${3:-0}
export TMPDIR=/tmp
D=/tmp/gmcli; rm -rf $D; mkdir -p $D/out
python - $WL $SC <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from genmap_amd import synth
codes, lens, desc = synth.workload(sys.argv[1], float(sys.argv[2]))
lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
off = 0
with open("/tmp/gmcli/genome.fa", "wb") as f:
    for k, ln in enumerate(lens):
        f.write(b">seq%d synthetic\n" % (k + 1))
        seq = lut[codes[off:off + ln]]; off += ln
        for i in range(0, len(seq), 60 * 100000):
            chunk = seq[i:i + 60 * 100000]
            n = len(chunk) // 60 * 60
            if n: f.write(b"\n".join(bytes(r) for r in chunk[:n].reshape(-1, 60)) + b"\n")
            if n < len(chunk): f.write(bytes(chunk[n:]) + b"\n")
print(desc)
PY
ls -la $D/genome.fa
TIMEFORMAT="genmap index wall: %R s"; time genmap_amd/bin/genmap index -F $D/genome.fa -I $D/idx -v 2>&1 | tail -4
du -sh $D/idx
TIMEFORMAT="genmap map e0 -r -fs wall: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 0 -r -fs -v -D $DEV 2>&1 | tail -5
cp $D/out/genome.genmap.freq8 $D/e0.freq8
TIMEFORMAT="genmap map e0 -bg -fs (GPU run-length form) wall: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 0 -bg -fs -v 2>&1 | tail -4
if [ "$WL" != grch38 ]; then
  TIMEFORMAT="genmap map e2 -r -fs wall: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 30 -E 2 -r -fs -v -D $DEV 2>&1 | tail -4
fi
ls -la $D/out | head; head -3 $D/out/genome.genmap.bedgraph
python - $WL $SC <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import genmap_amd as g
from genmap_amd import synth
codes, lens, desc = synth.workload(sys.argv[1], float(sys.argv[2]))
ix = g.Index.build(codes, lens, sampling=1)
E = 0 if sys.argv[1] == "grch38" else 2
ref = ix.map(30, E, value_bits=8)
got = np.fromfile("/tmp/gmcli/out/genome.genmap.freq8", dtype=np.uint8)
print(f"CLI freq8 (E={E}) == library:", np.array_equal(ref, got), len(got))
if E != 0:
    print("CLI freq8 (E=0) == library:", np.array_equal(ix.map(30, 0, value_bits=8), np.fromfile("/tmp/gmcli/e0.freq8", dtype=np.uint8)))
PY
