#!/bin/bash
# One kernel trace of one command: tools/gpu_prof_cmd.sh TAG python3 tools/x.py args...  -> gpurun_out/prof_TAG/kernel_stats.{csv,txt}
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT/prof_$TAG
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o t -- "$@" > $OUT/${TAG}_prof.log 2> $OUT/${TAG}_prof.err
echo "== prof rc=$?"
db=$(find $OUT/prof_$TAG -name '*_results.db' | head -n 1)
python3 tools/kernel_trace_summary.py "$db" $OUT/prof_$TAG/kernel_stats.csv > $OUT/prof_$TAG/kernel_stats.txt 2>&1
head -n 30 $OUT/prof_$TAG/kernel_stats.txt
rm -f "$db"
exit 0
