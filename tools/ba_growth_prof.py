"""dev tool: one scene of tools/ba_growth.py (N key frames, default 52) solved 60 times -- run under rocprofv3 for the kernel split."""
import _env  # noqa: F401
import sys

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 52
ctx = Context(0)
w = ba_workload(n_cams=n, n_points=1200, visibility=0.3, seed=n)
args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
for _ in range(60):
    g = ctx.ba_solve(*args)
print(n, "key frames:", g["trials"], "trials", ctx.ba_last_path())
ctx.close()
