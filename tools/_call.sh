cd $GRAFT_REPO_ROOT
(timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 --durations=15 2>&1 | tail -40) > gpurun_out/c12_pytest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c12_smoke.txt 2>&1
(timeout 900 python bench.py 2> gpurun_out/c12_bench.err > gpurun_out/c12_bench.json)
