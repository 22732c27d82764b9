cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests/ -x -q -m gpu --timeout 1200 2>&1 | tail -6) > gpurun_out/c55_pytest_gpu_full.txt
(timeout 600 python bench.py --workload bacteria5 --steps 10 2>/dev/null | tail -1) > gpurun_out/c55_bench_c5.json
