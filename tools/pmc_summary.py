"""Summarise rocprofv3 --pmc runs (FETCH_SIZE / WRITE_SIZE passes) of bench.py into per-kernel HBM traffic per forward.

Usage: python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <forwards profiled> > profiles/rNN_pmc_traffic.json
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KB; on gfx950 FETCH_SIZE reports half of a wide
coalesced read stream, so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16-byte stores.
"""
import collections, csv, glob, json, sys

root, n_fwd = sys.argv[1], int(sys.argv[2])
out = {"forwards": n_fwd, "unit": "bytes per forward", "correction": "fetch = 2 x FETCH_SIZE KB x 1024 (gfx950), write = WRITE_SIZE KB x 1024", "kernels": {}}
for name, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    f = glob.glob(f"{root}/pmc_{name}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(float)
    big = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        v = float(r["Counter_Value"]) * 1024.0 * mult
        acc[k] += v
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 400000:  # the eight conv layers, not the small GEMMs
            big[k] += v
    for k, v in acc.items():
        out["kernels"].setdefault(k, {})[name.lower().replace("_size", "_bytes")] = v / n_fwd
    for k, v in big.items():
        out["kernels"].setdefault(k + " [dispatches > 0.4 ms]", {})[name.lower().replace("_size", "_bytes")] = v / n_fwd
print(json.dumps(out, indent=1))
