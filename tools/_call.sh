cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 100 -k "baseline_settings_small or sixteen_symbol or gtest_matrix or correction_pass_near or reference_fixture or midsize" > gpurun_out/r04/c5_pytest_kind1.txt 2>&1; tail -5 gpurun_out/r04/c5_pytest_kind1.txt) || exit 1
grep -q passed gpurun_out/r04/c5_pytest_kind1.txt || exit 1
grep -q failed gpurun_out/r04/c5_pytest_kind1.txt && exit 1
(timeout 400 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,2,0.03 30,1,0.2 100,1,0.5 -- "" "jump_groups=1" "jump_groups=0" > gpurun_out/r04/c5_sweep_kind1.txt 2>&1; cat gpurun_out/r04/c5_sweep_kind1.txt)
(timeout 300 python tools/stats_run.py --workload grch38 --cfg 30,2 30,1 --frac 0.03 --settings "" > gpurun_out/r04/c5_stats_kind1.txt 2>&1; tail -2 gpurun_out/r04/c5_stats_kind1.txt | cut -c1-1700)
