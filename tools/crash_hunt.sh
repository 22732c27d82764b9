#!/bin/bash
# Hunt for the intermittent crash of `genmap index` (round 4: 1 SIGSEGV in ~350 process starts, case 3c, `-A skew -S 10`).
# Loops the index commands of the fixture cases 3a-3f (-S 10, -S 3, full array) N times, P processes at a time, first with the
# product binary (its crash handler prints the faulting stack), then with the AddressSanitizer + UBSan build of the host code
# (genmap_amd/host: make asan).  Every run that does not exit 0 is kept with its stderr.
#   tools/crash_hunt.sh [runs-per-binary, default 2000] [parallel, default 8] [outdir, default gpurun_out/crash_hunt]
N=${1:-2000}; P=${2:-8}; O=${3:-gpurun_out/crash_hunt}
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
mkdir -p $O; W=$(mktemp -d /tmp/crash_hunt.XXXX)
[ -x genmap_amd/bin/genmap_asan ] || make -C genmap_amd/host asan > /dev/null
for c in 3a 3b 3c 3d 3e 3f; do mkdir -p $W/fa_$c; cp tests/golden/reference_cases/case_$c/*.fa $W/fa_$c/ 2>/dev/null; done
one() {   # $1 binary  $2 run number
  local bin=$1 i=$2 cases=(3a 3b 3c 3d 3e 3f) samp=("-S 10" "-S 3" "" "-S 10" "-S 10" "-S 2")
  local c=${cases[$((i % 6))]} s=${samp[$(((i / 6) % 6))]} d=$W/idx_$(basename $bin)_$i
  local src; if ls $W/fa_$c/genome.fa > /dev/null 2>&1 && [ $(ls $W/fa_$c | wc -l) = 1 ]; then src="-F $W/fa_$c/genome.fa"; else src="-FD $W/fa_$c"; fi
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 $bin index $src -I $d -A skew $s > $d.out 2> $d.err; local rc=$?
  if [ $rc != 0 ]; then echo "run $i case $c $s: exit $rc"; { echo "== $(basename $bin) run $i case $c '$s' exit $rc"; tail -40 $d.err; } >> $O/failures_$(basename $bin).txt; fi
  rm -rf $d $d.out $d.err
}
export -f one; export W O
for bin in genmap_amd/bin/genmap genmap_amd/bin/genmap_asan; do
  : > $O/failures_$(basename $bin).txt
  t0=$(date +%s)
  seq 0 $((N - 1)) | xargs -P $P -I{} bash -c "one $ROOT/$bin {}" > $O/log_$(basename $bin).txt 2>&1
  echo "$(basename $bin): $N runs in $(( $(date +%s) - t0 )) s, failures: $(grep -c '^== ' $O/failures_$(basename $bin).txt)" | tee -a $O/summary.txt
done
rm -rf $W
