#!/bin/bash
# instruction volume of the search kernel per wave iteration (instruction-issue is what bounds e >= 1)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
SC=${SC:-0.2}
run() { name=$1; ctrs=$2; shift 2
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d gpurun_out/pmc/$name -o p --output-format csv -- python bench.py "$@" --scale $SC --no-cpu-baseline --no-counters --warmup 0 --steps 1 > gpurun_out/pmc/$name.log 2>&1
  python - "$name" <<'PY'
import csv,glob,sys,collections
name=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'gpurun_out/pmc/{name}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'search_kernel' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
for c,d in acc.items():
    vals=list(d.values()); print(name, c, 'dispatches', len(vals), 'last %.6g' % vals[-1])
PY
  rm -rf gpurun_out/pmc/$name
}
for E in 0 2; do
  run i${E}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" --E $E
  run i${E}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" --E $E
done
python tools/stats_run.py --scale $SC --cfg 30,0 30,2 2>&1 | tail -2 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('E',d['E'],'wave_iterations',d['wave_iterations'],'lanes_active',round(d['lanes_active_per_iteration'],1),'steps/kmer',round(d['steps_per_kmer'],2),'fetch/verify/step', round(d['cyc_fetch_frac'],3), round(d['cyc_verify_frac'],3), round(d['cyc_step_frac'],3), 'cyc/iter', round(d['cyc_per_iter']))"
