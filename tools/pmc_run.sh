#!/bin/bash
# PMC passes for the search kernel (each counter group in its own run, --kernel-trace only)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() { # name, counters, bench args
  name=$1; ctrs=$2; shift 2
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d gpurun_out/pmc/$name -o p --output-format csv -- python bench.py "$@" --no-cpu-baseline --no-counters --warmup 0 --steps 1 > gpurun_out/pmc/$name.log 2>&1
  python - "$name" <<'PY'
import csv,glob,sys,collections
name=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(f'gpurun_out/pmc/{name}/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'search_kernel' not in k: continue
        acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
for c,d in acc.items():
    vals=list(d.values())
    print(name, c, 'dispatches', len(vals), 'mean %.6g' % (sum(vals)/len(vals)), 'last %.6g' % vals[-1])
PY
}
for E in 0 2; do
  run e${E}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" --E $E
  run e${E}_b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" --E $E
  run e${E}_c "FETCH_SIZE" --E $E
  run e${E}_d "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" --E $E
done
