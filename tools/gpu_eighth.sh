#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer %.1f" % rf.get("node_steps_per_kmer"), "frac %.3f" % rf["frac"])'
echo "== pytest gpu all"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== stats"; timeout 900 python tools/stats_run.py 2>&1 | grep -v amdgpu
echo "== occupancy with verification, E=2 / E=0"
for pc in 4 6 8; do echo -n "perCU=$pc E=2: "; GM_BLOCKS_PER_CU=$pc timeout 900 python bench.py --E 2 --steps 1 --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4g' % r['value'], r['roofline']['kernel_ms'])"; done
for pc in 4 6 8; do echo -n "perCU=$pc E=0: "; GM_BLOCKS_PER_CU=$pc timeout 900 python bench.py --E 0 --steps 3 --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4g' % r['value'], r['roofline']['kernel_ms'])"; done
echo "== grch38 e0, e2"
timeout 1800 python bench.py --workload grch38 --steps 2 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_grch38_e0_v.json | python -c "$J"
timeout 2400 python bench.py --workload grch38 --E 2 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_grch38_e2_v.json | python -c "$J"
