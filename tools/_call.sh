cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q --timeout 200 -k "gtest_matrix or sampled_suffix or exclude_pseudo or five_bacteria or builder" > gpurun_out/r04/c11_pytest.txt 2>&1; tail -3 gpurun_out/r04/c11_pytest.txt)
grep -q passed gpurun_out/r04/c11_pytest.txt || exit 1
grep -q failed gpurun_out/r04/c11_pytest.txt && exit 1
(timeout 500 python bench.py --workload bacteria5 --no-cpu-baseline > gpurun_out/r04/c11_bench_c5.json 2> gpurun_out/r04/c11_bench_c5.log; python -c "
import json; d=json.loads(open('gpurun_out/r04/c11_bench_c5.json').read().strip().splitlines()[-1]); print('ep ms', d['ms_per_step'], 'csv', d['csv']['ms_per_pass'], d['csv']['locate_ms'])")
(timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/c11_prof -o p --output-format csv -- python bench.py --workload bacteria5 --no-cpu-baseline --no-counters --steps 1 --warmup 0 > gpurun_out/r04/c11_prof.log 2>&1; python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r04/c11_prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    out = open('gpurun_out/r04/c11_c5_kernel_stats.txt', 'w')
    for r in rows[:25]:
        line = f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} total ms {int(r['TotalDurationNs'])/1e6:10.2f} avg ms {float(r['AverageNs'])/1e6:9.3f} {r['Percentage']}%"
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/r04/c11_prof)
(timeout 600 python tools/wide_rows_smoke.py > gpurun_out/r04/c11_wide_rows_smoke.txt 2>&1; tail -12 gpurun_out/r04/c11_wide_rows_smoke.txt)
