"""dev tool: host phases of vs_ba_solve (VS_BA_TIMING=1, stderr) on the bundle-adjustment problems of the real-sequence fixtures
(tests/golden/real_ba_*.npz: what the driver's key frames hand to the solver) and wall time per call."""
import os
os.environ["VS_BA_TIMING"] = "1"
import _env  # noqa: F401,E402
import statistics  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

from visual_slam_amd import Context  # noqa: E402

ctx = Context(0)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("early", "middle", "last"):
    f = np.load(os.path.join(root, "real_ba_%s.npz" % name))
    args = (f["poses"], f["pose_fixed"], f["points"], f["point_fixed"], f["obs_pose"], f["obs_point"], f["obs_uv"], tuple(f["K"]))
    kw = dict(huber_delta=float(f["huber_delta"]), max_iterations=10, dcs_phi=float(f["dcs_phi"]),
              scale_edges=(f["scale_parent"].tolist(), f["scale_child"].tolist(), f["scale_meas"].tolist()))
    for _ in range(3):
        g = ctx.ba_solve(*args, **kw)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        g = ctx.ba_solve(*args, **kw)
        ts.append(time.perf_counter() - t0)
    sys.stderr.flush()
    print("real_ba_%s: %d poses, %d points, %d observations: %.1f us per ba_solve call (median of 15), %d trials, path %s" % (
        name, len(f["poses"]), len(f["points"]), len(f["obs_pose"]), statistics.median(ts) * 1e6, g["trials"], ctx.ba_last_path()), flush=True)
ctx.close()
