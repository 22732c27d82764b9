#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=150
timeout 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search" > $O/pytest_split.txt 2>&1; tail -3 $O/pytest_split.txt
grep -q "passed" $O/pytest_split.txt || exit 1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 30,1,0.1 --reps 1 -- "expand=0" "expand=1,expand_overlap=0" "expand=1" "expand=1,expand_occ=2" "expand=1,blocks_per_cu=4" "expand=1,fetch_batch=16" "expand=1,expand_overlap=0,blocks_per_cu=3" > $GRAFT_REPO_ROOT/$O/ab.txt 2>&1
cd $GRAFT_REPO_ROOT
grep "K=" $O/ab.txt
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/r06f/prof/*/*_kernel_trace.csv')[0])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows if ('expand_kernel' in r['Kernel_Name'] or 'CountEnv' in r['Kernel_Name'] or 'finalize' in r['Kernel_Name'])]
ev.sort()
call=[]
for s,e,n in ev:
    call.append((s,e,n))
    if 'finalize' in n:
        A=sum(e-s for s,e,n in call if 'expand_kernel' in n)/1e6
        B=sum(e-s for s,e,n in call if 'CountEnv<1, 2>' in n)/1e6
        J=sum(e-s for s,e,n in call if 'CountEnv<1, 1>' in n)/1e6
        print(f"slices={sum(1 for c in call if 'expand_kernel' in c[2]):3d} A={A:8.2f} B={B:8.2f} oneloop={J:8.2f} span={(call[-1][1]-call[0][0])/1e6:8.2f}")
        call=[]
PY
