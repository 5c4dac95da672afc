"""Summarise rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE passes) of bench.py into per-kernel HBM traffic per forward.

Usage: python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <forwards profiled> > profiles/rNN_pmc_traffic.json
Reads rocprofv3's rocpd sqlite output (`*_results.db`, view counters_collection) or the older counter_collection.csv.
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KB; on gfx950 FETCH_SIZE reports half of a wide
coalesced read stream, so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte stores.
"""
import collections, csv, glob, json, os, sqlite3, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odevio_amd._lib import source_sha  # noqa: E402


def rows(root, name):
    dbs = glob.glob(f"{root}/pmc_{name}/**/*_results.db", recursive=True)
    if dbs:
        c = sqlite3.connect(dbs[0])
        for k, v, s, e in c.execute("select kernel_name, value, start, end from counters_collection where counter_name = ?", (name,)):
            yield k, float(v), int(e) - int(s)
        return
    f = glob.glob(f"{root}/pmc_{name}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            yield r["Kernel_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


root, n_fwd = sys.argv[1], int(sys.argv[2])
out = {"source_sha": source_sha(), "forwards": n_fwd, "unit": "bytes per forward (and per WORKING dispatch: dispatches lasting > 10 % of the kernel's longest one - "
       "launches that return at once, like the predicated kernels of the CDE solver, are left out of that average)",
       "correction": "fetch = 2 x FETCH_SIZE KB x 1024 (gfx950), write = WRITE_SIZE KB x 1024", "kernels": {}}
for name, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    per = collections.defaultdict(list)
    for kname, value, dur in rows(root, name):
        k = kname.split("(")[0].replace("void ", "")
        per[k].append((value * 1024.0 * mult, dur))
    key = name.lower().replace("_size", "_bytes")
    for k, lst in per.items():
        e = out["kernels"].setdefault(k, {})
        e[key] = sum(v for v, _ in lst) / n_fwd
        longest = max(d for _, d in lst)
        work = [(v, d) for v, d in lst if d > 0.1 * longest]
        e[key + "_per_working_dispatch"] = sum(v for v, _ in work) / len(work)
        e["working_dispatches"] = len(work)
        e["avg_working_ns"] = round(sum(d for _, d in work) / len(work), 1)
print(json.dumps(out, indent=1))
