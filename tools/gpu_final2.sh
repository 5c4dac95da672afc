#!/bin/bash
# Round-end procedure, third gpurun call: phase stamps of the integrator and the conv kernel on the FINAL sources (diagnostic build
# make STAMPS=1), the training-step timings and their kernel trace, the bench lines of the other BASELINE configurations.
set -u
TAG=${1:-final}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
bash tools/gpu_run.sh ${TAG}x \
  "integ_stamps|200|python tools/integrator_stamps.py" \
  "conv_stamps|300|python tools/conv_stamps.py" \
  "train_full|200|python tools/time_train_step.py full" \
  "train_frozen|120|python tools/time_train_step.py frozen" \
  "time_backward|120|python tools/time_backward.py" \
  "time_backward_dopri5_gru|200|python tools/time_backward.py dopri5 gru" \
  "time_cde_backward|400|python tools/time_cde_backward.py 1024" \
  "bench_line|300|python bench.py" \
  "forward_sizes|300|python tools/time_forward_sizes.py" \
  "bench_dopri5|200|python bench.py --ode-solver dopri5 --drop 0.5 --no-cpu-baseline --no-f32-reference" \
  "bench_fp16|200|python bench.py --dtype fp16 --no-cpu-baseline" || exit $?
mkdir -p $OUT/prof_${TAG}train
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}train -o t -- python3 $ROOT/tools/time_train_step.py frozen > $OUT/${TAG}x_train_prof.log 2> $OUT/${TAG}x_train_prof.err
echo "== train_prof rc=$?"
db=$(find $OUT/prof_${TAG}train -name '*_results.db' | head -n 1)
python3 tools/kernel_trace_summary.py "$db" $OUT/prof_${TAG}train/kernel_stats.csv > $OUT/prof_${TAG}train/kernel_stats.txt 2>&1
head -n 12 $OUT/prof_${TAG}train/kernel_stats.txt
exit 0
