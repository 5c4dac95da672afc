#!/bin/bash
# CDE parity tests + timings in one gpurun call.  usage: tools/gpu_cde.sh TAG
set -u
TAG=${1:-cde}
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "cde" > $OUT/${TAG}_tests.log 2>&1
rc=$?; tail -n 15 $OUT/${TAG}_tests.log
[ $rc -gt 1 ] && exit $rc
for args in "1024 16 dopri5 1.0" "1024 16 dopri5 0.0" "512 16 dopri5 1.0" "128 16 dopri5 1.0"; do
  timeout -k 10 300 python tools/time_cde.py $args > $OUT/${TAG}_time_$(echo $args | tr ' ' '_').log 2>&1
  rc=$?; tail -n 4 $OUT/${TAG}_time_$(echo $args | tr ' ' '_').log
  [ $rc -gt 1 ] && exit $rc
done
exit 0
