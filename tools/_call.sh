cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 300 python -m pytest tests/test_gpu_parity.py -x -q --timeout 100 -k "baseline_settings_small or sixteen_symbol or gtest_matrix" > gpurun_out/r04/c9_pytest.txt 2>&1; tail -3 gpurun_out/r04/c9_pytest.txt)
grep -q passed gpurun_out/r04/c9_pytest.txt || exit 1
grep -q failed gpurun_out/r04/c9_pytest.txt && exit 1
(timeout 500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,2,0.03 30,1,0.2 100,1,0.5 -- "" "verify_t_ext=1" "lds_stack=3" "lds_stack=1" > gpurun_out/r04/c9_sweep.txt 2>&1; cat gpurun_out/r04/c9_sweep.txt)
