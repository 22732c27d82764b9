cd $GRAFT_REPO_ROOT
(timeout 1200 python tools/wide_rows_smoke.py 0.7 14,15,16 2>&1 | grep -v amdgpu.ids | tail -14) > gpurun_out/c46_wide.txt
