cd $GRAFT_REPO_ROOT
(timeout 1200 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 30,1,0.2 30,2,0.03 100,1,0.5 -- "" "prefetch_rec=0" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c17_sweep_pf.txt
(timeout 600 bash tools/cli_c5_check.sh 2>&1 | tail -40) > gpurun_out/c17_cli_c5.txt
