#!/bin/bash
# tools/profile_round.sh without the SQ counter passes and the step statistics (the first three parts: ~16 GPU-minutes)
#   1. the default bench line (3.09 Gbp, K=30 e=0 + sub-records)                          -> bench_default.json
#   2. rocprofv3 --kernel-trace --stats of the same command (headline launch only)        -> kernel_stats_short.csv
#   3. the four configurations (30,0) (30,1) (30,2) (100,1) in ONE process under rocprofv3: kernel durations per configuration,
#      and PMC traffic FETCH_SIZE / WRITE_SIZE in separate passes                            -> kernel_by_config.txt, pmc_by_config.txt
#   4. SQ counters + device-side step statistics of the search kernels                     -> pmc_sq_grch38.txt, step_stats_grch38.txt
R=${1:-r03}; O=gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
CFGS="30,0,1.0 30,1,1.0 30,2,1.0 100,1,1.0"
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
echo "== the four configurations in one process: kernel durations (rocprofv3 --kernel-trace)"
# sweep_tuning.py runs reps + 1 = 3 calls per configuration in the order of --cfg; a call with errors on this Dna5 text is a
# search_kernel (CountEnv) dispatch followed by the small correction dispatch (ScatterEnv)
timeout 1500 rocprofv3 --kernel-trace -d $O/kt -o p --output-format csv -- python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg $CFGS -- "" > $O/kt.log 2>&1
python - $O "$CFGS" <<'PY'
import csv, glob, sys
O, cfgs = sys.argv[1], sys.argv[2].split()
rows = []
for f in glob.glob(f'{O}/kt/**/*kernel_trace.csv', recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if 'search_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
main = [r for r in rows if 'ScatterEnv' not in r['Kernel_Name']]
corr = [r for r in rows if 'ScatterEnv' in r['Kernel_Name']]
with open(f'{O}/kernel_by_config.txt', 'w') as out:
    out.write('# rocprofv3 --kernel-trace of tools/sweep_tuning.py --reps 2 (3 calls per configuration, the first one untimed warm-up); ms per dispatch\n')
    for k, c in enumerate(cfgs):
        d = main[3 * k:3 * k + 3]
        ms = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in d]
        name = d[0]['Kernel_Name'] if d else '?'
        out.write(f'cfg {c}: {name[:70]}  dispatches ms {[round(x, 3) for x in ms]}  mean of the timed two {sum(ms[1:]) / max(1, len(ms) - 1):.3f}\n')
    cms = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in corr]
    out.write(f'correction pass (ScatterEnv) dispatches: {len(cms)}, ms {[round(x, 3) for x in cms]}\n')
print(open(f'{O}/kernel_by_config.txt').read())
PY
rm -rf $O/kt
echo "== PMC traffic per configuration (FETCH_SIZE, WRITE_SIZE in separate passes)"
: > $O/pmc_by_config.txt
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 1500 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p --output-format csv -- python tools/sweep_tuning.py --workload grch38 --reps 1 --cfg $CFGS -- "" > $O/pmc_$C.log 2>&1
  python - $C $O "$CFGS" <<'PY'
import csv, glob, sys, collections
c, O, cfgs = sys.argv[1], sys.argv[2], sys.argv[3].split()
acc = collections.OrderedDict()
for f in glob.glob(f'{O}/pmc_{c}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search_kernel' in r['Kernel_Name'] and 'ScatterEnv' not in r['Kernel_Name'] and r['Counter_Name'] == c:
            acc[int(r['Dispatch_Id'])] = acc.get(int(r['Dispatch_Id']), 0.0) + float(r['Counter_Value'])
vals = [acc[d] for d in sorted(acc)]
with open(f'{O}/pmc_by_config.txt', 'a') as out:
    for k, cf in enumerate(cfgs):   # 2 dispatches per configuration: the second one
        v = vals[2 * k + 1] if 2 * k + 1 < len(vals) else float('nan')
        out.write(f'{c} cfg {cf}: {v:.6g} KB per search_kernel dispatch = {v * 1024 / 1e9:.3f} GB (64 B per request)\n')
print(open(f'{O}/pmc_by_config.txt').read())
PY
  rm -rf $O/pmc_$C
done
