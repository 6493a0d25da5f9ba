#!/bin/bash
# dev aid: kernel stats of the scaled bundle adjustment under rocprofv3 for a library variant (VS_LIB_PATH) -> gpurun_out/prof_scaled_<tag>/
#   bash tools/prof_scaled_lib.sh <tag> [ba_scaled.py arguments]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/prof_scaled_$tag
rm -rf $O && mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o scaled -- python3 $R/tools/ba_scaled.py "$@" > $O/log.txt 2>&1 || exit 1
echo "== $tag"
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/scaled_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), "%10.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
grep -h "GPU:\|max rel" $O/log.txt
