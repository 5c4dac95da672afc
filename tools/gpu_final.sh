#!/bin/bash
# Round-end procedure, as TWO gpurun calls (each stays inside the 1200 s limit):
#   1. gpurun -- "bash tools/gpu_suite.sh TAG"       GPU suite (production library), the same suite against the bounds-audit build, bench
#   2. gpurun -- "bash tools/gpu_final.sh TAG"       this script: rocprofv3 passes of the bench (kernel trace + stats; FETCH_SIZE, WRITE_SIZE and
#      the utilisation counters in passes of their own), the Neural-CDE bench lines and their passes
# then copy gpurun_out/prof_TAG*/{kernel_stats.csv,pmc_traffic.json,pmc_*_derived.json} and the bench lines into profiles/rNN_*
# (bench.py only quotes a committed PMC file whose source_sha matches the sources it runs).
set -u
TAG=${1:-final}
DERIVE="conv_f16x2 conv1_f16x2 integrator" bash tools/gpu_profile.sh $TAG 13 "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_BUSY_avr TA_BUSY_max" -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-reference || exit $?
bash tools/gpu_run.sh ${TAG}c "cde|500|python bench.py --model cde" "cdebf|300|python bench.py --model cde --dtype bf16 --no-cpu-baseline" || exit $?
bash tools/gpu_profile.sh ${TAG}cde 4 -- --model cde --steps 3 --warmup 1 --no-cpu-baseline
