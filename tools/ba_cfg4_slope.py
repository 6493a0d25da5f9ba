"""dev tool: cfg4 solve time against the number of LM iterations (intercept = host structure pass + upload + read-back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload
ctx = Context(0)
w = ba_workload()
args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
for it in (0, 1, 2, 5, 10):
    for _ in range(3): r = ctx.ba_solve(*args, max_iterations=it)
    t0 = time.perf_counter()
    for _ in range(10): r = ctx.ba_solve(*args, max_iterations=it)
    dt = (time.perf_counter() - t0) / 10
    print("max_iterations %2d: %.1f us per solve (%d trials)" % (it, dt * 1e6, r["trials"]))
