#!/bin/bash
# rocprofv3 passes of one bench.py command line in one gpurun call: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE
# in passes of their own (counters are never combined with a trace).  Results under gpurun_out/prof_TAG/; summaries
# (the files to commit under profiles/) are written next to them.
#   usage: tools/gpu_profile.sh TAG FORWARDS [extra pmc pass ...] -- bench.py arguments
set -u
TAG=$1; NFWD=$2; shift 2
EXTRA=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do EXTRA+=("$1"); shift; done
shift
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
run() {  # name, command...
  local name=$1; shift
  timeout -k 10 600 "$@" > $OUT/$name.log 2> $OUT/$name.err
  local rc=$?
  echo "== $name rc=$rc"; tail -n 2 $OUT/$name.log | cut -c1-400
  if [ $rc -gt 1 ]; then tail -n 20 $OUT/$name.err; exit $rc; fi
}
run trace rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py "$@"
run fetch rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_FETCH_SIZE -o p -- python3 $ROOT/bench.py "$@"
run write rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_WRITE_SIZE -o p -- python3 $ROOT/bench.py "$@"
for grp in "${EXTRA[@]}"; do
  run "pmc_$(echo $grp | cut -c1-24 | tr ' ' '_')" rocprofv3 --pmc $grp -d "$OUT/pmc/$(echo $grp | tr ' ' '_' | cut -c1-40)" -o p -- python3 $ROOT/bench.py "$@"
done
db=$(find $OUT/trace -name '*_results.db' | head -n 1)
python3 $ROOT/tools/kernel_trace_summary.py "$db" $OUT/kernel_stats.csv > $OUT/kernel_stats.txt 2>&1
python3 $ROOT/tools/pmc_summary.py $OUT $NFWD > $OUT/pmc_traffic.json 2> $OUT/pmc_summary.err
# derived utilisation figures (clock, MFMA busy, LDS, TA, wave wait) per kernel named in $DERIVE, from the extra passes
for k in ${DERIVE:-}; do
  python3 $ROOT/tools/pmc_derive.py $OUT/pmc $k 100 > $OUT/pmc_${k}_derived.json 2>> $OUT/pmc_summary.err
done
cat $OUT/kernel_stats.txt | head -n 14
exit 0
