set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_match.py -x -q -m gpu > gpurun_out/r4_match_tests.log 2>&1 || { tail -30 gpurun_out/r4_match_tests.log; exit 1; }
tail -1 gpurun_out/r4_match_tests.log
TSTAGE=1 timeout -k 10 200 python tools/match_stamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_match_stamps.log
grep -E "tstage|cycle counter|tail|words arrived|own scan|all waves scan" gpurun_out/r4_match_stamps.log
for i in 1 2; do
VS_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_prev.so TSTAGE=1 BLOCKS=0 timeout -k 10 200 python tools/match_sweep.py 2>&1 | grep tstage | sed 's/^/prev /'
TSTAGE=1 BLOCKS=0 timeout -k 10 300 python tools/match_sweep.py 2>&1 | grep tstage | sed 's/^/new  /'
done
VS_LIB_PATH=$GRAFT_REPO_ROOT/build/lib_prev.so TSTAGE=1 BLOCKS=0 timeout -k 10 200 python tools/match_sweep.py 100000 100000 2>&1 | grep tstage | sed 's/^/prev /'
TSTAGE=1 BLOCKS=0 timeout -k 10 300 python tools/match_sweep.py 100000 100000 2>&1 | grep tstage | sed 's/^/new  /'
