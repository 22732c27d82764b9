cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "baseline_settings_small or sixteen_symbol or gtest_matrix or correction_pass_near or reference_fixture" > gpurun_out/r04/c4_pytest_filter2.txt 2>&1; tail -5 gpurun_out/r04/c4_pytest_filter2.txt)
(timeout 900 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,2,0.03 30,1,0.2 100,1,0.5 -- "" "jump_filter=2" "jump_filter=2,jump_groups=0" "verify_t=2" > gpurun_out/r04/c4_sweep_filter2.txt 2>&1; cat gpurun_out/r04/c4_sweep_filter2.txt)
(timeout 600 python tools/stats_run.py --workload grch38 --cfg 30,2 30,1 --frac 0.03 --settings "" "jump_filter=2" > gpurun_out/r04/c4_stats_filter2.txt 2>&1; tail -4 gpurun_out/r04/c4_stats_filter2.txt | cut -c1-1700)
