#!/bin/bash
O=gpurun_out/r06p; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=200
python - <<'PY' > $O/small_share.txt 2>&1
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, genmap_amd as g, helpers as H
rng = np.random.default_rng(7)
lens = [70000, 700, 29, 40000, 3, 30]
codes = rng.integers(0, 4, size=sum(lens), dtype=np.uint8)
fam = rng.integers(0, 4, size=300, dtype=np.uint8)
for s in rng.integers(0, 100000, size=40):
    cp = fam.copy(); m = rng.random(300) < 0.03; cp[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8); codes[s:s + 300] = cp
codes[30000:30500] = 4; codes[rng.integers(0, len(codes), 30)] = 4
ora = H.OracleIndex(codes, lens, keep_sa=False)
ix = g.Index.build(codes, lens, sampling=1)
ok = True
for K, E in ((30, 1), (30, 2), (24, 2), (50, 3), (36, 4)):
    exp = ora.mappability(K, E, value_bits=16, threads=8)
    for two in (0, 1):
        for mb in (-1, 1):
            ix.set_tuning(expand=1, expand_share=1, expand_two_pass=two, expand_mb=mb)
            out = ix.map(K, E, value_bits=16)
            same = np.array_equal(out, exp); ok = ok and same
            print(K, E, "two_pass", two, "mb", mb, "ok" if same else ("DIFFERS at %s" % np.flatnonzero(out != exp)[:8]), flush=True)
    iv = [(100, 9000), (60000, 75000)]
    ix.set_tuning(expand=1, expand_share=1, expand_two_pass=1, expand_mb=2)
    same = np.array_equal(ix.map(K, E, value_bits=16, intervals=iv), ora.mappability(K, E, value_bits=16, intervals=iv, threads=8)); ok = ok and same
    print(K, E, "selection", "ok" if same else "DIFFERS", flush=True)
print("ALL OK" if ok else "FAILED")
PY
tail -30 $O/small_share.txt
grep -q "ALL OK" $O/small_share.txt || exit 1
timeout 900 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.1 30,1,0.3 --reps 1 -- "expand_share=0" "expand_share=1" "expand_share=1,expand_occ=8" > $O/ab.txt 2>&1; grep "K=" $O/ab.txt
