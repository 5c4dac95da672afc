#!/bin/bash
# Kernel trace of one command, then the durations of every dispatch of kernels matching PATTERN in the LAST n dispatches:
#   tools/gpu_trace_seq.sh TAG PATTERN N python3 tools/x.py args...
set -u
TAG=$1; PAT=$2; N=$3; shift 3
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT -o t -- "$@" > $OUT/cmd.log 2> $OUT/cmd.err
echo "== trace rc=$?"
db=$(find $OUT -name '*_results.db' | head -n 1)
python3 - "$db" "$PAT" "$N" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, end-start from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)))
for r in rows[-int(sys.argv[3]):]:
    print(f"  {r[0][:40]:40s} grid {r[1] // r[4]:6d} x {r[2]:3d} x {r[3]:3d}  {r[5] / 1e3:9.1f} us")
PY
rm -f "$db"
exit 0
