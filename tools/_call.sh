cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 900 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,2,0.03 -- "" "STEP=5" "STEP=7" "STEP=8" "probation=0" "probation=2" "steal=8" "steal=32" "steal=0" "fetch_batch=8" "fetch_batch=16" "oss_weights=34629" "oss_weights=34886" "sat_min_w=128" "verify_cost=2" "verify_cost=4" > gpurun_out/r04/c13_sweep_retune_e2.txt 2>&1; cat gpurun_out/r04/c13_sweep_retune_e2.txt)
(timeout 600 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,1,0.2 -- "" "STEP=6" "STEP=10" "STEP=12" "probation=1" "probation=3" "fetch_batch=16" "fetch_batch=64" "steal=4" "verify_t_ext=8" > gpurun_out/r04/c13_sweep_retune_e1.txt 2>&1; cat gpurun_out/r04/c13_sweep_retune_e1.txt)
