cd $GRAFT_REPO_ROOT
(timeout 2400 python -m pytest tests/ -x -q -m gpu --timeout 1200 2>&1 | tail -8) > gpurun_out/c37_pytest_gpu_full.txt
(timeout 600 python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/c37_smoke.txt
