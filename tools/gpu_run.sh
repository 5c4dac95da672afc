#!/bin/bash
# Generic step runner for one gpurun call: each argument is "name|timeout|command"; a step that is killed or times out
# (rc > 1) ends the call.  Logs go to gpurun_out/TAG_name.log / .err.
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; t=${rest%%|*}; cmd=${rest#*|}
  echo "== $name: $cmd"
  timeout -k 10 $t bash -c "$cmd" > $OUT/${TAG}_$name.log 2> $OUT/${TAG}_$name.err
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 $OUT/${TAG}_$name.log | cut -c1-600
  if [ $rc -ne 0 ]; then tail -n 12 $OUT/${TAG}_$name.err | cut -c1-400; fi
  if [ $rc -gt 1 ]; then echo "step $name ended abnormally: stopping"; exit $rc; fi
done
exit 0
