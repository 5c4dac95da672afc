#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace result database (rocpd sqlite): per-kernel stats CSV and, optionally, the
dispatch sequence of the last forward.  Usage: kernel_trace_summary.py results.db [out.csv] [--last-forward]"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = next((a for a in sys.argv[2:] if not a.startswith("--")), None)
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    # WORKING dispatches: those lasting > 10 % of the kernel's longest one.  Kernels that are launched predicated (the CDE
    # solver's stages return at once when the controller does not want them) would otherwise average in their no-ops.
    work = {}
    for r in rows:
        d = [x[0] for x in c.execute("select end-start from kernels where name = ? and (end-start) > ?", (r[0], 0.1 * r[5]))]
        work[r[0]] = (len(d), sum(d) / len(d))
    if out:
        w = csv.writer(open(out, "w"))
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "WorkingCalls", "WorkingAverageNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5], work[r[0]][0], round(work[r[0]][1], 1)])
    for r in rows[:12]:
        print(f"{r[0][:60]:60s} calls {r[1]:5d} avg {r[3] / 1e3:10.1f} us  {100 * r[2] / tot:6.2f} %   working {work[r[0]][0]:5d} avg {work[r[0]][1] / 1e3:10.1f} us")
    if "--last-forward" in sys.argv:
        seq = list(c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, end-start from kernels order by start"))
        idx = [i for i, r in enumerate(seq) if r[0].startswith("ingest") or r[0].startswith("conv1_kernel")]
        for r in seq[idx[-1]:idx[-1] + 32]:
            print(f"  {r[0][:48]:48s} grid {r[1] // r[4]:6d} x {r[2]:3d} x {r[3]:3d}  {r[5] / 1e3:9.1f} us")


if __name__ == "__main__":
    main()
