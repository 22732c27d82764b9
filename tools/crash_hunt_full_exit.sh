#!/bin/bash
# the rare SIGSEGV of `genmap` under the GPU suite's conditions, WITHOUT the fast exit (GENMAP_FULL_EXIT=1: static destructors and the HIP
# runtime's own teardown run), core dumps on.  The suite's pattern: one `genmap index` of a fixture directory with -S 10, then six `genmap map` processes at a
# time on a GPU that other processes have just left.   tools/crash_hunt_full_exit.sh [rounds, default 100] [outdir]
N=${1:-100}; O=${2:-gpurun_out/crash_hunt_r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
mkdir -p $O; W=$(mktemp -d /tmp/crash6.XXXX); ulimit -c unlimited 2>/dev/null
export GENMAP_FULL_EXIT=1
B=$ROOT/genmap_amd/bin/genmap
mkdir -p $W/fa; cp tests/golden/reference_cases/case_3a/*.fa $W/fa/
fails=0; starts=0
: > $O/failures.txt
for i in $(seq 1 $N); do
  rm -rf $W/idx; $B index -FD $W/fa -I $W/idx -A skew -S 10 > $W/i.out 2> $W/i.err; rc=$?; starts=$((starts + 1))
  if [ $rc != 0 ]; then fails=$((fails + 1)); { echo "== round $i index exit $rc"; tail -60 $W/i.err; } >> $O/failures.txt; continue; fi
  pids=()
  for j in 1 2 3 4 5 6; do
    mkdir -p $W/o$j; ( $B map -I $W/idx -O $W/o$j -K 4 -E $((j % 2)) -r -fl $([ $((j % 3)) = 0 ] && echo -ep) > $W/m$j.out 2> $W/m$j.err; echo $? > $W/m$j.rc ) &
    pids+=($!)
  done
  wait "${pids[@]}"
  for j in 1 2 3 4 5 6; do
    starts=$((starts + 1)); rc=$(cat $W/m$j.rc)
    if [ "$rc" != 0 ]; then fails=$((fails + 1)); { echo "== round $i map $j exit $rc"; tail -60 $W/m$j.err; } >> $O/failures.txt; fi
    rm -rf $W/o$j
  done
done
echo "full exit (GENMAP_FULL_EXIT=1): $starts process starts ($N rounds of one index + six concurrent maps), failures: $fails; cores: $(ls core* $W/core* 2>/dev/null | wc -l)" | tee $O/summary.txt
rm -rf $W
