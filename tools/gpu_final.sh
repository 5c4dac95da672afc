#!/bin/bash
# Round-end run in one gpurun call: GPU suite (production library), the same suite against the bounds-audit build, the
# bench line, then the rocprofv3 passes of the bench (kernel trace + stats, FETCH_SIZE, WRITE_SIZE and the utilisation
# counters in passes of their own).  usage: tools/gpu_final.sh TAG
set -u
TAG=${1:-final}
bash tools/gpu_run.sh $TAG \
  "tests|900|python -m pytest tests -q -m gpu -p no:cacheprovider" \
  "audit|900|ODEVIO_LIB=\$PWD/odevio_amd/libodevio_audit.so python -m pytest tests -q -m gpu -p no:cacheprovider -k 'not cde_hidden_1024 and not backward'" \
  "bench|400|python bench.py" || exit $?
DERIVE="conv_f16x2 conv1_f16x2 integrator" bash tools/gpu_profile.sh $TAG 13 "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_BUSY_avr TA_BUSY_max" -- --steps 10 --warmup 3 --no-cpu-baseline --no-f32-reference
