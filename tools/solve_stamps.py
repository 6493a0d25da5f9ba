"""dev tool: phase cycles of ba_solve_block (a build with -DVS_SOLVE_STAMPS, loaded through VS_LIB_PATH) for the third LM trial of
one solve of each real-sequence fixture and of cfg4."""
import _env  # noqa: F401
import os

import numpy as np

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

ctx = Context(0)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("early", "middle", "last"):
    f = np.load(os.path.join(root, "real_ba_%s.npz" % name))
    for _ in range(3):
        ctx.ba_solve(f["poses"], f["pose_fixed"], f["points"], f["point_fixed"], f["obs_pose"], f["obs_point"], f["obs_uv"], tuple(f["K"]),
                     huber_delta=float(f["huber_delta"]), dcs_phi=float(f["dcs_phi"]),
                     scale_edges=(f["scale_parent"].tolist(), f["scale_child"].tolist(), f["scale_meas"].tolist()))
w = ba_workload()
for _ in range(3):
    ctx.ba_solve(w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
ctx.synchronize()
ctx.close()
