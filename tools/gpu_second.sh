#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== gather sizes"; timeout 600 tools/gather_bench 64 sweep > gpurun_out/gather2.txt 2>&1; tail -40 gpurun_out/gather2.txt
echo "== rocprof chr1 e0"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_chr1_e0 -o r01 --output-format csv -- python bench.py --steps 3 --no-cpu-baseline > gpurun_out/prof_chr1_e0.txt 2>&1; tail -3 gpurun_out/prof_chr1_e0.txt
find gpurun_out/prof_chr1_e0 -name "*stats*" | head
for f in $(find gpurun_out/prof_chr1_e0 -name "*kernel_stats.csv"); do head -12 $f; done
echo "== block bytes sweep chr1 e0"
for bb in 32 64 128; do timeout 600 python bench.py --steps 3 --block-bytes $bb --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print($bb, r['value'], r['roofline']['kernel_ms'], r['roofline'].get('rank_lines'))"; done
echo "== occupancy sweep chr1 e0"
for pc in 2 4 8; do GM_BLOCKS_PER_CU=$pc timeout 600 python bench.py --steps 3 --no-cpu-baseline --no-counters 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print($pc, r['value'], r['roofline']['kernel_ms'])"; done
echo "== chr1 e1, e2, k100e1"
timeout 900 python bench.py --steps 2 --E 1 --no-cpu-baseline > gpurun_out/bench_chr1_e1.txt 2>&1; tail -1 gpurun_out/bench_chr1_e1.txt | cut -c1-900
timeout 1200 python bench.py --steps 1 --E 2 --no-cpu-baseline > gpurun_out/bench_chr1_e2.txt 2>&1; tail -1 gpurun_out/bench_chr1_e2.txt | cut -c1-900
timeout 900 python bench.py --steps 2 --K 100 --E 1 --no-cpu-baseline > gpurun_out/bench_chr1_k100e1.txt 2>&1; tail -1 gpurun_out/bench_chr1_k100e1.txt | cut -c1-900
echo "== grch38 e0"
timeout 1800 python bench.py --workload grch38 --steps 2 > gpurun_out/bench_grch38_e0.txt 2>&1; tail -6 gpurun_out/bench_grch38_e0.txt | cut -c1-1500
