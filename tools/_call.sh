cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 400 -k "sampled or 64_bit_rows or gtest or baseline_settings_small or fixture or midsize" 2>&1 | tail -8) > gpurun_out/c15_pytest.txt
(timeout 1200 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 30,1,0.2 30,2,0.03 100,1,0.5 -- "" "verify2=0" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c15_sweep_v2.txt
