#!/bin/bash
# One gpurun call: the GPU test-suite against the production library, then against the bounds-audit build, then the
# bench line.  A step that is killed or times out (rc > 1) ends the call: no further GPU step is started.
#   usage: tools/gpu_suite.sh TAG [pytest -k expression for the audit pass]
set -u
TAG=${1:-run}
KAUDIT=${2:-not cde_hidden_1024}
OUT=gpurun_out
mkdir -p $OUT
step() {  # name, timeout seconds, command...
  local name=$1 t=$2; shift 2
  echo "== $name" | tee -a $OUT/${TAG}_steps.log
  timeout -k 10 $t "$@" > $OUT/${TAG}_$name.log 2> $OUT/${TAG}_$name.err
  local rc=$?
  echo "== $name rc=$rc" | tee -a $OUT/${TAG}_steps.log
  tail -n 3 $OUT/${TAG}_$name.log
  if [ $rc -gt 1 ]; then echo "step $name ended abnormally (rc=$rc): stopping"; tail -n 20 $OUT/${TAG}_$name.err; exit $rc; fi
  return $rc
}
step tests 900 python -m pytest tests -q -m gpu -p no:cacheprovider
step audit 900 env ODEVIO_LIB=$PWD/odevio_amd/libodevio_audit.so python -m pytest tests -q -m gpu -p no:cacheprovider -k "$KAUDIT"
step bench 600 python bench.py
step bench_again 300 python bench.py --no-cpu-baseline --no-f32-reference   # a second line: run-to-run spread on the same box
exit 0
