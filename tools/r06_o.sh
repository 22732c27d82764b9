#!/bin/bash
O=gpurun_out/r06o; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=300
timeout 700 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "map_files or five_bacteria" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python bench.py --workload bacteria5 --no-traffic > $O/bench_c5.json 2> $O/bench_c5.log; tail -c 900 $O/bench_c5.json; tail -3 $O/bench_c5.log
timeout 600 python bench.py --workload bacteria5 --no-traffic --c5-per-file --no-cpu-baseline --no-csv > $O/bench_c5_per_file.json 2>/dev/null; python -c "
import json;a=json.load(open('$O/bench_c5.json'));b=json.load(open('$O/bench_c5_per_file.json'));print('one launch',a['ms_per_step'],a['value'],'csv',a.get('csv',{}).get('ms_per_pass') if a.get('csv') else None,'| per file',b['ms_per_step'],b['value'])"
