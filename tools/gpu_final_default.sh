#!/bin/bash
# the default bench line, the rocprofv3 kernel statistics of the same command and the PMC traffic, for profiles/
mkdir -p gpurun_out/r01h; export TMPDIR=/tmp; O=gpurun_out/r01h
timeout 1500 python bench.py > $O/bench_chr1_e0.json 2> $O/bench_chr1_e0.log; tail -1 $O/bench_chr1_e0.json | cut -c1-1500
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof -o r01h --output-format csv -- python bench.py --no-cpu-baseline > $O/prof.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r01h/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    out=open('gpurun_out/r01h/kernel_stats_short.csv','w'); w=csv.writer(out); w.writerow(['Name','Calls','TotalDurationNs','AverageNs','Percentage'])
    for r in rows:
        n=r['Name']; n=n if len(n)<90 else n[:60]+'...'+n[-25:]
        w.writerow([n,r['Calls'],r['TotalDurationNs'],r['AverageNs'],r['Percentage']])
    out.close()
    print(open('gpurun_out/r01h/kernel_stats_short.csv').read()[:900])
PY
rm -rf $O/prof
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p --output-format csv -- python bench.py --no-cpu-baseline --no-counters --steps 2 > $O/pmc_$C.log 2>&1
  python - $C <<'PY'
import csv,glob,sys
c=sys.argv[1]; vals=[]
for f in glob.glob(f'gpurun_out/r01h/pmc_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search_kernel' in r['Kernel_Name'] and r['Counter_Name']==c: vals.append(float(r['Counter_Value']))
print(c, 'per search_kernel dispatch (KB):', vals)
open(f'gpurun_out/r01h/pmc_{c}.txt','w').write(f'{c} per search_kernel dispatch (KB): {vals}\n')
PY
  rm -rf $O/pmc_$C
done
