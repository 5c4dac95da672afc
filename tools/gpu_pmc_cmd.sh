#!/bin/bash
# Utilisation counters of one kernel of one command: tools/gpu_pmc_cmd.sh TAG KERNEL python3 tools/x.py args...
# (five passes of the command, one counter group each; never combined with a trace) -> gpurun_out/prof_TAG/pmc_KERNEL_derived.json
set -u
TAG=$1; KERNEL=$2; shift 2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "TA_BUSY_avr TA_BUSY_max" "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/pmc/$(echo $grp | tr ' ' '_' | cut -c1-40)"
  timeout -k 10 300 rocprofv3 --pmc $grp -d "$d" -o p -- "$@" > $OUT/pmc.log 2> $OUT/pmc.err
  rc=$?
  echo "== $grp rc=$rc"
  if [ $rc -gt 1 ]; then tail -n 10 $OUT/pmc.err; exit $rc; fi
done
python3 tools/pmc_derive.py $OUT/pmc $KERNEL 100 > $OUT/pmc_${KERNEL}_derived.json 2> $OUT/derive.err
cat $OUT/pmc_${KERNEL}_derived.json | head -60
find $OUT/pmc -name '*_results.db' -delete
exit 0
