set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_r5mc
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --memory-copy-trace --hip-runtime-trace -d $OUT -o t -- python3 tools/time_train_step.py frozen > $OUT/cmd.log 2> $OUT/cmd.err
echo rc=$?
db=$(find $OUT -name '*_results.db' | head -n 1)
python3 - "$db" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'memory' in t.lower()][:10])
try:
    rows = list(c.execute("select name, count(*), sum(size), sum(end-start) from memory_copies group by name"))
    for r in rows: print(r)
    rows = list(c.execute("select size, count(*) from memory_copies where name like '%DEVICE_TO_DEVICE%' group by size order by 2 desc limit 25"))
    for r in rows: print('D2D size', r)
except Exception as e:
    print('err', e)
try:
    rows = list(c.execute("select name, count(*) from regions where name like 'hipMemcpy%' or name like 'hipMemset%' group by name order by 2 desc limit 12"))
    for r in rows: print(r)
except Exception as e:
    print('err2', e)
PY
rm -f "$db"
