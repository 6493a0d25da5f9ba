#!/bin/bash
# dev aid: kernel stats of the scaled bundle adjustment under rocprofv3 -> gpurun_out/prof_scaled/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_scaled
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o scaled -- python3 $R/tools/ba_scaled.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/scaled_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), "%10.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
