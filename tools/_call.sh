cd $GRAFT_REPO_ROOT
(timeout 900 bash tools/cli_c5_check.sh 1.0 0 2>&1 | grep -v amdgpu.ids | tail -45) > gpurun_out/c58_cli_c5.txt
(timeout 1200 python bench.py 2> gpurun_out/c58_bench_default.log | tail -1) > gpurun_out/c58_bench_default.json
