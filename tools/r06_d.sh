#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 --reps 1 -- "expand=1,sat_draw_w=1,expand_occ=1" "expand=1,sat_draw_w=1,expand_occ=2" "expand=1,sat_draw_w=1,expand_occ=4" "expand=1,sat_draw_w=1,expand_occ=6" "expand=1,sat_draw_w=1,expand_occ=8" "expand=1,sat_draw_w=1,expand_occ=16" "expand=1,sat_draw_w=1,expand_chunk=4" "expand=1,sat_draw_w=1,expand_chunk=72" > $GRAFT_REPO_ROOT/$O/ab.txt 2>&1
cd $GRAFT_REPO_ROOT
grep "K=" $O/ab.txt
python - <<'PY'
import csv,glob
rows=list(csv.DictReader(open(glob.glob('gpurun_out/r06d/prof/*/*_kernel_trace.csv')[0])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Grid_Size_X') or r.get('Grid_Size')) for r in rows if ('expand_kernel' in r['Kernel_Name'] or 'CountEnv' in r['Kernel_Name'] or 'finalize' in r['Kernel_Name'])]
ev.sort()
call=[]
for s,e,n,g in ev:
    call.append((s,e,n,g))
    if 'finalize' in n:
        A=sum(e-s for s,e,n,g in call if 'expand_kernel' in n)/1e6
        B=sum(e-s for s,e,n,g in call if 'CountEnv<1, 2>' in n)/1e6
        gs=[g for s,e,n,g in call if 'expand_kernel' in n][:1]
        print(f"slices={sum(1 for c in call if 'expand_kernel' in c[2]):3d} A={A:8.2f} B={B:8.2f} gridA={gs}")
        call=[]
PY
