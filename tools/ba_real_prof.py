"""dev tool: the real-sequence bundle-adjustment fixtures (tests/golden/real_ba_<name>.npz) solved 120 times -- run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of what the driver's key frames hand to the solver.
    python tools/ba_real_prof.py [early|middle|last]"""
import _env  # noqa: F401
import os
import statistics
import sys
import time

import numpy as np

from visual_slam_amd import Context

name = sys.argv[1] if len(sys.argv) > 1 else "last"
ctx = Context(0)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
f = np.load(os.path.join(root, "real_ba_%s.npz" % name))
args = (f["poses"], f["pose_fixed"], f["points"], f["point_fixed"], f["obs_pose"], f["obs_point"], f["obs_uv"], tuple(f["K"]))
kw = dict(huber_delta=float(f["huber_delta"]), max_iterations=10, dcs_phi=float(f["dcs_phi"]),
          scale_edges=(f["scale_parent"].tolist(), f["scale_child"].tolist(), f["scale_meas"].tolist()))
ts = []
for _ in range(120):
    t0 = time.perf_counter()
    g = ctx.ba_solve(*args, **kw)
    ts.append(time.perf_counter() - t0)
print("real_ba_%s: %d poses, %d points, %d observations: %.1f us per solve (median of the last 50), %d trials -> %.1f us per trial; %s" % (
    name, len(f["poses"]), len(f["points"]), len(f["obs_pose"]), statistics.median(ts[-50:]) * 1e6, g["trials"],
    statistics.median(ts[-50:]) * 1e6 / g["trials"], ctx.ba_last_path()))
ctx.close()
