set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05_realba
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in ${@:-early middle last}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- python3 $R/tools/ba_real_prof.py $n > $O/${n}_under_rocprof.log 2>&1
  python3 $R/tools/trace_by_grid.py $O/$n/${n}_kernel_trace.csv > $O/real_ba_${n}_by_grid.csv
  rm -f $O/$n/${n}_kernel_trace.csv $O/$n/*agent_info.csv
  tail -1 $O/${n}_under_rocprof.log
done
