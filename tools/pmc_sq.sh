# dev tool: SQ counters of the match kernel in both train-staging modes (separate rocprofv3 --pmc passes, counters only)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq_r4
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for TS in 0 1; do
  export TSTAGE=$TS
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/ts$TS -o p -- python3 $R/tools/match_run.py > $O/ts$TS.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/ts${TS}b -o p -- python3 $R/tools/match_run.py > $O/ts${TS}b.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, os, collections
root = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_sq_r4")
for d in sorted(glob.glob(os.path.join(root, "ts*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "hamming_knn2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())}, "dispatches", max((len(v) for v in acc.values()), default=0))
PY
find $O -name "*counter_collection.csv" -delete
