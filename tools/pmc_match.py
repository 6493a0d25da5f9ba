"""Turns the counter CSVs of three separate rocprofv3 --pmc passes over the 10k x 10k match into profiles/rNN_pmc_match.json.
Passes (each its own process, counters only, no trace domains):
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/fetch -o p -- python3 bench.py --no-frames --no-cpu-baseline --steps 20 --warmup 5
  rocprofv3 --pmc WRITE_SIZE ...                 -d out/write
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE ... -d out/sq
usage: python tools/pmc_match.py out > profiles/r03_pmc_match.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
grids = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for k in ("hamming_knn2_kernel",):
            if k in name:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                grids[k].add(r.get("Grid_Size", "?"))
counters = {k: {c: {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "dispatches": len(v)} for c, v in cs.items()}
            for k, cs in acc.items()}
p = counters["hamming_knn2_kernel"]
fetch_kb, write_kb = p["FETCH_SIZE"]["mean"], p["WRITE_SIZE"]["mean"]
nq = nt = 10000
out = {
    "command": __doc__.split("usage")[0].strip(),
    "workload": "cfg3 10000 x 10000 x 256-bit, 1 GPU (MI355X)",
    "grid_sizes_seen": {k: sorted(v) for k, v in grids.items()},
    "counters": counters,
    "hamming_knn2_kernel_per_launch": {
        "FETCH_SIZE_KB_raw": fetch_kb, "fetch_bytes_raw": fetch_kb * 1024,
        "fetch_bytes_gfx950_x2_upper_bound": 2 * fetch_kb * 1024,
        "WRITE_SIZE_KB": write_kb, "write_bytes": write_kb * 1024,
        "hbm_bytes_raw": (fetch_kb + write_kb) * 1024,
        "hbm_bytes_corrected_upper": (2 * fetch_kb + write_kb) * 1024,
        "algorithmic_bytes_streamed_model": 32.0 * nq * nt,
        "compulsory_bytes": 32 * (nq + nt) + int(write_kb * 1024),
        "note": "FETCH_SIZE on gfx950 under-reports wide coalesced streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM); "
                "this kernel mixes 16-B/lane query loads with coalesced 32-B/lane train-row loads that are staged through "
                "LDS (64 rows per wave at a time), an uncalibrated pattern, so both the raw value and the 2x upper bound are "
                "given.  WRITE_SIZE = the per-chunk partial rows "
                "(31 publishing chunks x 40 tiles x 256 x 8 B, write-through) + the 160 KB of results; the folding workgroups poll those words (their loads bypass the caches).",
    },
}
if "SQ_INSTS_VALU" in p:
    out["valu"] = {"SQ_INSTS_VALU_per_launch": p["SQ_INSTS_VALU"]["mean"],
                   "expected_18.5_ops_x_1e8_pairs_div_64": 18.5 * 1e8 / 64,
                   "per_SIMD": p["SQ_INSTS_VALU"]["mean"] / 1024}
json.dump(out, sys.stdout, indent=1)
