set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --single-stream --steps 20 --warmup 5 > $O/bench_under_rocprof.log 2>&1
echo "bench profiled"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scaled -o scaled -- python3 $R/tools/ba_scaled.py > $O/ba_scaled_under_rocprof.log 2>&1
echo "scaled profiled"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/resident -o resident -- python3 $R/tools/resident_prof.py > $O/resident_under_rocprof.log 2>&1
echo "resident profiled"
cd $R
python3 tools/trace_by_grid.py $O/bench/bench_kernel_trace.csv > $O/bench_by_grid.csv
python3 tools/trace_by_grid.py $O/resident/resident_kernel_trace.csv > $O/resident_by_grid.csv
timeout -k 10 300 python3 tools/ba_scaled.py --check > $O/ba_scaled.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o cfg4 -- python3 $R/tools/ba_cfg4_prof.py > $O/cfg4_under_rocprof.log 2>&1
python3 tools/trace_by_grid.py $O/cfg4/cfg4_kernel_trace.csv > $O/cfg4_by_grid.csv
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
python3 tools/trace_timeline.py $O/resident/resident_kernel_trace.csv 0.90 70 > $O/tracking_timeline.txt
python3 tools/pnp_stamps.py > $O/pnp_stamps.txt 2>&1
python3 tools/mo_stamps.py > $O/mo_stamps.txt 2>&1
python3 tools/match_stamps.py > $O/match_stamps.txt 2>&1
python3 tools/chain_gaps.py $O/resident/resident_kernel_trace.csv > $O/tracking_chain.txt 2>&1
python3 tools/api_stages.py 2>&1 | grep -v Warning > $O/class_api_stages.txt
python3 tools/ba_window_check.py --big > $O/ba_window_check.txt 2>&1
rm -f $O/*/*_kernel_trace.csv $O/*/*agent_info.csv
