#!/bin/bash
# first contact with the MI355X: microbenchmark, parity tests, small and full bench
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export TMPDIR=/tmp
rocm-smi --showmeminfo vram 2>&1 | head -8 > gpurun_out/smi.txt
nproc >> gpurun_out/smi.txt; lscpu | grep -E "Model name|Socket|Core" >> gpurun_out/smi.txt; free -g | head -2 >> gpurun_out/smi.txt
echo "== gather"; timeout 300 tools/gather_bench 16 > gpurun_out/gather.txt 2>&1; tail -50 gpurun_out/gather.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -5 gpurun_out/smoke.txt
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -25 gpurun_out/pytest_gpu.txt
echo "== bench small"; timeout 600 python bench.py --scale 0.1 --steps 3 > gpurun_out/bench_small.txt 2>&1; tail -12 gpurun_out/bench_small.txt
echo "== bench full"; timeout 1200 python bench.py --steps 3 > gpurun_out/bench_full.txt 2>&1; tail -12 gpurun_out/bench_full.txt
