cd $GRAFT_REPO_ROOT
(timeout 1500 bash tools/cli_scale_check.sh grch38 1.0 0 2>&1 | grep -v amdgpu.ids | tail -40) > gpurun_out/c47_cli_scale.txt
