#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "bb", r["config"]["block_bytes"], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "lines", rf.get("rank_lines"), "steps/kmer", rf.get("node_steps_per_kmer"), "frac", rf.get("frac"))'
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== occupancy sweep chr1 e0 (32B)"
for pc in 1 2 3 4 6; do echo -n "perCU $pc: "; GM_BLOCKS_PER_CU=$pc timeout 600 python bench.py --steps 3 --no-cpu-baseline --no-counters 2>/dev/null | python -c "$J"; done
echo "== chr1 e0/e1/e2/k100e1 defaults"
for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do timeout 1200 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== chr1 e1 64B"; timeout 900 python bench.py --E 1 --steps 2 --block-bytes 64 --no-cpu-baseline 2>/dev/null | python -c "$J"
echo "== grch38 e0 32B / 64B / 128B"
for bb in 32 64 128; do timeout 1800 python bench.py --workload grch38 --steps 2 --block-bytes $bb --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== pmc calibration on gather"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_gather -o g --output-format csv -- tools/gather_bench 4 > gpurun_out/pmc_gather.txt 2>&1
ls gpurun_out/pmc_gather; head -3 gpurun_out/pmc_gather/*counter_collection.csv
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pmc_gather/*counter_collection.csv'):
    rows=list(csv.DictReader(open(f)))
    print(rows[0].keys())
    for r in rows:
        if 'gather_kernel' in r.get('Kernel_Name',''):
            print(r.get('Kernel_Name','')[:40], r.get('Grid_Size'), r.get('Counter_Name'), r.get('Counter_Value'))
PY
