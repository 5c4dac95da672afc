#!/usr/bin/env python
"""Sum every counter of rocprofv3 --pmc passes per kernel (largest dispatches only with --min-us).
Usage: pmc_generic.py <dir> [--min-us 400]   (walks <dir> for *counter_collection.csv)"""
import collections, csv, glob, sys

root = sys.argv[1]
min_ns = 0
if "--min-us" in sys.argv:
    min_ns = float(sys.argv[sys.argv.index("--min-us") + 1]) * 1e3
for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < min_ns:
            continue
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    print("==", f)
    for k in sorted(acc):
        print(f"  {k[0]:40s} {k[1]:34s} sum {acc[k]:.4e}  per dispatch {acc[k] / cnt[k]:.4e}  (n={cnt[k]})")
