#!/usr/bin/env python
"""Derived utilisation figures of one kernel from rocprofv3 --pmc passes (rocpd sqlite, one db per pass).

Usage: pmc_derive.py <dir with the pass directories> <kernel substring> [min dispatch us] > profiles/rNN_pmc_<kernel>_derived.json
  clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / kernel time          (the counter sums the XCDs)
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)
  lds_active     = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE / 8);  lds_conflict likewise
  ta_busy        = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8) per dispatch
  wave_wait      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
Each pass is a separate run of the same command, so ratios across passes assume the runs are alike (they are to ~2 %)."""
import glob, json, os, sqlite3, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odevio_amd._lib import source_sha  # noqa: E402

root, key = sys.argv[1], sys.argv[2]
min_ns = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0
c = {}
for f in sorted(glob.glob(f"{root}/**/*_results.db", recursive=True)):
    db = sqlite3.connect(f)
    q = ("select counter_name, count(*), sum(value), sum(end-start) from counters_collection "
         "where kernel_name like ? and (end-start) > ? group by counter_name")
    for name, n, total, dur in db.execute(q, (f"%{key}%", min_ns)):
        c[name] = {"n": n, "sum": total, "ns": dur}
out = {"source_sha": source_sha(), "kernel": key, "counters": c}
if "GRBM_GUI_ACTIVE" in c:
    g = c["GRBM_GUI_ACTIVE"]
    cyc = g["sum"] / 8.0
    out["clock_ghz"] = round(cyc / g["ns"], 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        out["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / 1024.0 / cyc, 3)
    per_ns = cyc / g["ns"]
    for nm, key2, div in (("lds_active", "SQ_LDS_IDX_ACTIVE", 256.0), ("lds_conflict", "SQ_LDS_BANK_CONFLICT", 256.0)):
        if key2 in c:
            out[nm] = round(c[key2]["sum"] / div / (c[key2]["ns"] * per_ns), 3)
    for nm, key2 in (("ta_busy_avg", "TA_BUSY_avr"), ("ta_busy_max", "TA_BUSY_max")):
        if key2 in c:
            out[nm] = round(c[key2]["sum"] / (c[key2]["ns"] * per_ns), 3)
if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
    out["wave_wait"] = round(c["SQ_WAIT_INST_ANY"]["sum"] / c["SQ_WAVE_CYCLES"]["sum"], 3)
print(json.dumps(out, indent=1))
