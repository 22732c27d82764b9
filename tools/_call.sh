cd $GRAFT_REPO_ROOT
run() { (timeout 300 python bench.py --workload bacteria5 --steps 5 --K 24 --E 1 --infix $1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('infix', $1, 'ep ms/step', round(d['ms_per_step'],1), 'csv pass ms', round(d['csv']['ms_per_pass'],1), 'locate', round(d['csv']['locate_ms'],1))") >> gpurun_out/c62_c5.txt 2>&1; }
run 0; run 20; run 17
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 600 -k "exclude or five_bacteria or fixture or csv or cli or locations or sampled" 2>&1 | tail -4) > gpurun_out/c62_pytest.txt
