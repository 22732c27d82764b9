cd $GRAFT_REPO_ROOT
(timeout 1500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 100,0,1.0 50,0,1.0 -- "" "verify_t=2" "verify_t=4" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c35_sweep.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 900 -k "sixteen or baseline_settings_small or fixture or chr1 or ecoli" 2>&1 | tail -5) > gpurun_out/c35_pytest.txt
