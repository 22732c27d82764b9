#!/bin/bash
# The measurement set that goes into profiles/r06/final (run on the GPU box): tools/profile_round6.sh [part ...]   parts: bench stats trace pmc sq steps
#   bench  the default bench line (3.09 Gbp, K=30 e=2 + sub-records, C2 and C5 on their own indexes)       -> bench_default.json
#   stats  rocprofv3 --kernel-trace --stats of the headline alone                                           -> kernel_stats_short.csv
#   trace  (30,0) (30,1) (30,2) (100,1) in ONE process under rocprofv3: per pass, ms of phase A / walker   -> kernel_by_config.txt
#   pmc    FETCH_SIZE / WRITE_SIZE of the same passes, separate runs                                        -> pmc_by_config.txt
#   sq     SQ counters of the same kernels                                                                   -> pmc_sq_grch38.txt
#   steps  device-side step statistics (instrumented twin)                                                   -> step_stats_grch38.txt
O=gpurun_out/r06final; mkdir -p $O
export TMPDIR=/tmp
PARTS=${@:-bench stats trace pmc sq steps}
CFGS="30,0,1.0 30,1,1.0 30,2,1.0 100,1,1.0"
for P in $PARTS; do case $P in
bench)
  echo "== bench default"; timeout 1700 python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.log;;
stats)
  echo "== rocprofv3 kernel stats (headline alone)"
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python bench.py --no-cpu-baseline --no-counters --no-host-rate --no-traffic --sub "" --extra-configs "" > $O/prof.log 2>&1
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
    print(open(f'{O}/kernel_stats_short.csv').read()[:1800])
PY
  rm -rf $O/prof;;
trace)
  echo "== kernel durations per pass"
  timeout 1500 rocprofv3 --kernel-trace -d $O/kt -o p --output-format csv -- python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg $CFGS -- "" > $O/kt.log 2>&1
  python tools/passes.py trace $(find $O/kt -name "*kernel_trace.csv" | head -1) $CFGS --reps 2 > $O/kernel_by_config.txt; cat $O/kernel_by_config.txt
  rm -rf $O/kt;;
pmc)
  echo "== PMC traffic per pass"
  : > $O/pmc_by_config.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 1500 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_$C -o p --output-format csv -- python tools/sweep_tuning.py --workload grch38 --reps 1 --cfg $CFGS -- "" > $O/pmc_$C.log 2>&1
    python tools/passes.py pmc $(find $O/pmc_$C -name "*counter_collection.csv" | head -1) $C $CFGS | sed 's/$/  (KB per pass; x 1024 = bytes, 64 B per request)/' >> $O/pmc_by_config.txt
    rm -rf $O/pmc_$C
  done
  cat $O/pmc_by_config.txt;;
sq)
  echo "== SQ counters"
  FR="30,0,1.0 30,1,0.2 100,1,1.0 30,2,0.06"
  : > $O/pmc_sq_grch38.txt
  for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE"; do
    timeout 900 rocprofv3 --pmc $G --kernel-trace -d $O/pmc_sq -o p --output-format csv -- python tools/sweep_tuning.py --workload grch38 --reps 1 --cfg $FR -- "" > $O/pmc_sq.log 2>&1
    for C in $G; do python tools/passes.py pmc $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $C $FR >> $O/pmc_sq_grch38.txt; done
    rm -rf $O/pmc_sq
  done
  cat $O/pmc_sq_grch38.txt;;
steps)
  echo "== step statistics"
  timeout 900 python tools/stats_run.py --workload grch38 --cfg 30,0 30,1 100,1 > $O/step_stats_grch38.txt 2>&1
  timeout 900 python tools/stats_run.py --workload grch38 --frac 0.1 --cfg 30,2 >> $O/step_stats_grch38.txt 2>&1
  cut -c1-400 $O/step_stats_grch38.txt;;
esac; done
ls -la $O
