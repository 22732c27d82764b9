cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 400 -k "sampled or csv_locations or exclude_pseudo_and_locations or 64_bit_rows" 2>&1 | tail -8) > gpurun_out/c14_pytest.txt
(timeout 900 python tools/sweep_tuning.py --workload grch38 --reps 2 --block-bytes 64 --cfg 30,0,1.0 30,1,0.2 30,2,0.03 100,1,0.5 -- "coop=1" "coop=0" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c14_sweep_bb64.txt
(timeout 600 python tools/stats_run.py --workload grch38 --cfg 30,0 --settings "coop=1" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c14_stats_bb32_e0.txt
