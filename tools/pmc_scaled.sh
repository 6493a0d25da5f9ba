#!/bin/bash
# dev aid: counter passes (separate processes, counters only) over the scaled bundle adjustment -> gpurun_out/pmc_scaled/summary.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_scaled
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/ba_scaled.py"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
cd $R && python3 - <<PY > $O/summary.txt
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join("$O", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
n_res = 2000000
print("# scaled bundle adjustment (100 cameras x 200000 points, 2 000 000 residuals): counters per launch (mean over the launches of two solves)")
print("# FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950 reports half of the bytes of wide coalesced streaming reads (MI355X_MICROARCH.md): bytes/residual given raw and with reads doubled")
print("%-22s %5s %12s %12s %9s %9s %12s %12s" % ("kernel", "n", "fetch KB", "write KB", "B/res raw", "B/res 2x", "VALU insts", "wave-cycles"))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
    m = lambda name: (sum(c[name]) / len(c[name])) if name in c else float("nan")
    f, w = m("FETCH_SIZE"), m("WRITE_SIZE")
    print("%-22s %5d %12.0f %12.0f %9.1f %9.1f %12.0f %12.0f" % (k[:22], len(c.get("FETCH_SIZE", [])), f, w, (f + w) * 1024 / n_res, (2 * f + w) * 1024 / n_res, m("SQ_INSTS_VALU"), m("SQ_WAVE_CYCLES")))
PY
find $O -name "*counter_collection.csv" -delete
cat $O/summary.txt
