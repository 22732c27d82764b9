cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1700 python bench.py > gpurun_out/r04/b1_bench_default.json 2> gpurun_out/r04/b1_bench_default.log; echo rc=$? >> gpurun_out/r04/b1_bench_default.log)
(timeout 600 python bench.py --workload bacteria5 > gpurun_out/r04/b1_bench_c5.json 2> gpurun_out/r04/b1_bench_c5.log; echo rc=$? >> gpurun_out/r04/b1_bench_c5.log)
