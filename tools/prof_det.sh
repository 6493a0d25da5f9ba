cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_det
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o det -- python3 $R/tools/detect_host_timing.py > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/det_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%8.1f us avg  %8.1f min" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
