#!/bin/bash
# The measurement set that goes into profiles/<round>/ (run on the GPU box):  tools/profile_round.sh r02
#   1. the default bench line (3.09 Gbp, K=30 e=0 + sub-records)          -> bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same command (headline only) -> kernel_stats_short.csv
#   3. PMC traffic of the same launch, separate passes                      -> pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt
#   4. SQ counters + device-side step statistics of the search kernel       -> pmc_sq_<workload>.txt, step_stats_<workload>.txt
R=${1:-r02}; O=gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
echo "== bench default"; timeout 1700 python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -c 1500 $O/bench_default.json; echo
echo "== rocprofv3 kernel stats (headline launch only)"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python bench.py --no-cpu-baseline --no-counters --no-host-rate --sub "" > $O/prof.log 2>&1
python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(f'{O}/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(f'{O}/kernel_stats_short.csv', 'w') as out:
        w = csv.writer(out); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
        for r in rows:
            n = r['Name']; n = n if len(n) < 90 else n[:60] + '...' + n[-25:]
            w.writerow([n, r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']])
    print(open(f'{O}/kernel_stats_short.csv').read()[:2500])
PY
rm -rf $O/prof
echo "== PMC traffic (FETCH_SIZE, WRITE_SIZE in separate passes)"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p --output-format csv -- python bench.py --no-cpu-baseline --no-counters --no-host-rate --sub "" --steps 2 --warmup 1 > $O/pmc_$C.log 2>&1
  python - $C $O <<'PY'
import csv, glob, sys
c, O = sys.argv[1], sys.argv[2]; vals = []
for f in glob.glob(f'{O}/pmc_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c: vals.append(float(r['Counter_Value']))
print(c, 'per search_kernel dispatch (KB):', vals)
open(f'{O}/pmc_{c}.txt', 'w').write(f'{c} per search_kernel dispatch (KB): {vals}\n')
PY
  rm -rf $O/pmc_$C
done
echo "== SQ counters of the search kernel (one group per pass) and device-side step statistics"
for WL in grch38 chr1; do
  FR="30,0,1.0 30,1,0.2 100,1,1.0 30,2,0.06"; [ $WL = chr1 ] && FR="30,0,1.0 30,1,1.0 100,1,1.0 30,2,0.5"
  : > $O/pmc_sq_$WL.txt
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE"; do
    timeout 900 rocprofv3 --pmc $G --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python tools/sweep_tuning.py --workload $WL --reps 1 --cfg $FR -- "" > $O/pmc_sq.log 2>&1
    python - $O $WL <<'PY'
import csv, glob, sys, collections
O, WL = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for f in glob.glob(f'{O}/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search_kernel' not in r['Kernel_Name']: continue
        acc.setdefault((int(r['Dispatch_Id']), r['Counter_Name']), 0.0)
        acc[(int(r['Dispatch_Id']), r['Counter_Name'])] += float(r['Counter_Value'])
disp = sorted({d for d, _ in acc})
with open(f'{O}/pmc_sq_{WL}.txt', 'a') as out:
    # sweep_tuning runs reps+1 = 2 dispatches per configuration, in the order of --cfg: take the second of each pair
    for k, d in enumerate(disp):
        if k % 2 == 1:
            for (dd, c), v in acc.items():
                if dd == d: out.write(f'cfg#{k // 2} {c} {v:.6g}\n')
PY
    rm -rf $O/pmc_sq
  done
  timeout 900 python tools/stats_run.py --workload $WL --cfg 30,0 30,1 100,1 $([ $WL = chr1 ] && echo 30,2) > $O/step_stats_$WL.txt 2>&1
done
ls -la $O
