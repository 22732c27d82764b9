cd $GRAFT_REPO_ROOT
(timeout 200 python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2) > gpurun_out/c63_smoke.txt
(timeout 120 python -m pytest tests/test_gpu_parity.py -x -q --timeout 100 -k "ecoli" 2>&1 | tail -2) >> gpurun_out/c63_smoke.txt
