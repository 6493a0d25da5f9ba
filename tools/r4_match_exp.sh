set -e
cd $GRAFT_REPO_ROOT
for ts in 3 4 5; do
python - <<PY
import sys; sys.path.insert(0,'tools'); import _env
import numpy as np
from visual_slam_amd import Context
from visual_slam_amd.workloads import match_workload
from oracle import oracle
ctx = Context(0); ctx.tune_match(tstage=$ts)
for nq, nt, seed in ((700, 900, 5), (1000, 257, 6), (3000, 3001, 7), (256, 64, 8), (513, 130, 9)):
    q, t = match_workload(nq, nt, n_dup=8, seed=seed)
    idx, dist = ctx.hamming_knn2(q, t); oi, od = oracle.hamming_knn2(q, t)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od), (nq, nt)
print("tstage $ts parity ok")
ctx.close()
PY
done
for i in 1 2; do
TSTAGE=2,3,4,5 BLOCKS=0,1024 timeout -k 10 300 python tools/match_sweep.py 2>&1 | grep tstage
done
TSTAGE=3,4,5 BLOCKS=0 timeout -k 10 300 python tools/match_sweep.py 100000 100000 2>&1 | grep tstage
