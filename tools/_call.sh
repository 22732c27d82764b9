cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --timeout 900 -k "fixture or gtest or ecoli or baseline_settings or midsize" 2>&1 | tail -5) > gpurun_out/r04/a5_parity.txt
(timeout 1500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 100,1,0.5 30,1,0.2 30,2,0.03 -- "" "verify_t_ext=2" "verify_t_ext=4" "verify_t_ext=8" "verify_t_ext=16" "steal=16,probation=0"  "range_add=0,verify_t_ext=1" 2>&1 | grep -v amdgpu.ids) > gpurun_out/r04/a5_sweep_vtext.txt
