set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/resident -o resident -- python3 $R/tools/resident_prof.py > $O/resident_under_rocprof.log 2>&1
cd $R
python3 tools/trace_by_grid.py $O/resident/resident_kernel_trace.csv > $O/resident_by_grid.csv
python3 tools/chain_gaps.py $O/resident/resident_kernel_trace.csv > $O/tracking_chain.txt 2>&1
python3 tools/trace_timeline.py $O/resident/resident_kernel_trace.csv 0.90 70 > $O/tracking_timeline.txt
cat $O/tracking_chain.txt
grep -E "pnp_ransac|ba_motion|detect_band|select_desc|hamming|ratio|track_" $O/resident_by_grid.csv | head -30
tail -5 $O/resident_under_rocprof.log
python3 tools/resident_prof.py 2>&1 | tail -4
rm -f $O/*/*_kernel_trace.csv $O/*/*agent_info.csv
