#!/bin/bash
# Times the bench's conv2..6 stage under a list of forced tile plans (ODEVIO_CONV_FORCE, api.hip plan_f16x2), one bench
# run each, same box: usage tools/gpu_plans.sh TAG "force1" "force2" ...   ("" = the planner's own choice)
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/${TAG}_plans.txt
for f in "$@"; do
  ODEVIO_CONV_FORCE="$f" timeout -k 10 150 python bench.py --no-cpu-baseline --no-f32-reference > $OUT/${TAG}_plan.log 2> $OUT/${TAG}_plan.err
  rc=$?
  python - "$f" $OUT/${TAG}_plan.log >> $OUT/${TAG}_plans.txt <<'PY'
import json, sys
l = [x for x in open(sys.argv[2]) if x.startswith("{")]
if l:
    d = json.loads(l[-1])
    print(f"{sys.argv[1] or '(planner)':28s} conv2_6 {d['stage_ms']['conv2_6']:.4f} ms  {d['value']:.0f} frames/s")
else:
    print(f"{sys.argv[1]:28s} FAILED")
PY
  tail -n 1 $OUT/${TAG}_plans.txt
  if [ $rc -gt 1 ]; then echo "bench ended abnormally: stopping"; exit $rc; fi
done
