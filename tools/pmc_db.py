#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc passes stored as rocpd sqlite (one db per pass).
Usage: pmc_db.py <dir> <kernel substring> [min duration us]"""
import glob, sqlite3, sys

root, key = sys.argv[1], sys.argv[2]
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0
for f in sorted(glob.glob(f"{root}/**/*_results.db", recursive=True)):
    c = sqlite3.connect(f)
    q = ("select counter_name, count(*), sum(value), sum(end-start) from counters_collection "
         "where kernel_name like ? and (end-start) > ? group by counter_name")
    for name, n, total, dur in c.execute(q, (f"%{key}%", min_ns)):
        print(f"{name:36s} n={n:4d} sum={total:.5e} per_dispatch={total / n:.5e} kernel_ns_sum={dur}")
