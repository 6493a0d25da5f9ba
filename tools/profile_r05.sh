set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --single-stream --steps 20 --warmup 5 > $O/bench_under_rocprof.log 2>&1
echo "bench profiled"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scaled -o scaled -- python3 $R/tools/ba_scaled.py > $O/ba_scaled_under_rocprof.log 2>&1
echo "scaled profiled"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o cfg4 -- python3 $R/tools/ba_cfg4_prof.py > $O/cfg4_under_rocprof.log 2>&1
cd $R
python3 tools/trace_by_grid.py $O/bench/bench_kernel_trace.csv > $O/bench_by_grid.csv
python3 tools/trace_by_grid.py $O/cfg4/cfg4_kernel_trace.csv > $O/cfg4_by_grid.csv
timeout -k 10 300 python3 tools/ba_scaled.py --check > $O/ba_scaled.log 2>&1
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log | cut -c1-200
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --force-collective --no-frames --no-cpu-baseline > $O/bench_force_collective.log 2> $O/bench_force_collective.err || echo "force-collective run failed"
tail -1 $O/bench_force_collective.log | cut -c1-200
python3 tools/keyframe_stages.py 4 20 > $O/keyframe_stages.txt 2>&1
python3 tools/ba_graph.py > $O/ba_graph.txt 2>&1
rm -f $O/*/*_kernel_trace.csv $O/*/*agent_info.csv
