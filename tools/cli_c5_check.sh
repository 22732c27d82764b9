#!/bin/bash
# BASELINE config C5 through the `genmap` program at its real size: five related genomes (~21 Mbp, 10 sequences) in a directory,
# `genmap index -FD`, `genmap map -K 24 -E 1 --exclude-pseudo --csv -r -fl`, timed (-v), counts checked against the library path.
#   tools/cli_c5_check.sh [scale devices]
SC=${1:-1.0}; DEV=${2:-0}
export TMPDIR=/tmp
D=/tmp/gmc5; rm -rf $D; mkdir -p $D/fa $D/out
python - $SC <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from genmap_amd import synth
lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
for name, recs in synth.bacteria5(float(sys.argv[1])):
    with open("/tmp/gmc5/fa/" + name, "wb") as f:
        for sname, codes in recs:
            f.write(b">" + sname.encode() + b"\n")
            seq = lut[codes]
            n = len(seq) // 70 * 70
            if n: f.write(b"\n".join(bytes(r) for r in seq[:n].reshape(-1, 70)) + b"\n")
            if n < len(seq): f.write(bytes(seq[n:]) + b"\n")
PY
ls -la $D/fa
TIMEFORMAT="genmap index -FD wall: %R s"; time genmap_amd/bin/genmap index -FD $D/fa -I $D/idx -v 2>&1 | tail -3
TIMEFORMAT="genmap map K24 E1 -ep -d -r -fl wall: %R s"; time genmap_amd/bin/genmap map -I $D/idx -O $D/out -K 24 -E 1 -ep -d -r -fl -v -D $DEV 2>&1 | grep -v "^- Index was\|^Index was\|BWT\|suffix array" | tail -22
ls -la $D/out | head -14
python - $SC <<'PY'
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import genmap_amd as g
from genmap_amd import synth
import helpers as H
files = synth.bacteria5(float(sys.argv[1]))
gen = H.Genome(files)
ix = g.Index.build(gen.codes, gen.seq_len, sampling=1)
ok = True
for name, first, nseq, tb, tl in gen.file_slices():
    ref = ix.map(24, 1, first_seq=first, n_seq=nseq, value_bits=16, exclude_pseudo=True, seq_file_id=gen.seq_file)
    got = np.fromfile("/tmp/gmc5/out/" + name.rsplit(".", 1)[0] + ".genmap.freq16", dtype=np.uint16)
    ok &= bool(np.array_equal(ref, got))
    rows = sum(1 for _ in open("/tmp/gmc5/out/" + name.rsplit(".", 1)[0] + ".genmap.csv")) - 1
    print(name, "freq16 == library:", np.array_equal(ref, got), "csv rows", rows, "positions with a hit", int((ref > 0).sum()))
print("C5_CLI_OK" if ok else "C5_CLI_FAIL")
PY
