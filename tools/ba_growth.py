"""dev tool: the cost of a growing global bundle adjustment (the reference's localBundleAdjustement is global: every key frame, every
point, LocalBA.py:143-172): scenes of 1 200 points seen from a random 30 % of N key frames, N = 10 .. 60, 10 LM iterations."""
import _env  # noqa: F401
import statistics
import time

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

import sys
ctx = Context(0)
if len(sys.argv) > 1:  # 1: systems beyond 126 unknowns through HBM (ba_chol_panel / ba_chol_update), as before the packed LDS solver
    ctx.tune_ba_solve(int(sys.argv[1]))
for n in (10, 16, 21, 22, 23, 27, 30, 34, 35, 40, 52, 60):
    w = ba_workload(n_cams=n, n_points=1200, visibility=0.3, seed=n)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    for _ in range(5):
        g = ctx.ba_solve(*args)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        g = ctx.ba_solve(*args)
        ts.append(time.perf_counter() - t0)
    p = ctx.ba_last_path()
    print("%2d key frames (%3d unknowns, %5d observations): %7.1f us per solve, %2d trials -> %6.1f us per trial; %s + %s" % (
        n, p["unknowns"], len(w["obs_pose"]), statistics.median(ts) * 1e6, g["trials"], statistics.median(ts) * 1e6 / max(g["trials"], 1),
        p["schur"], p["dense"]), flush=True)
ctx.close()
