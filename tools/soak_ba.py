"""dev tool: soak test of the general BA path on BASELINE cfg4 (last-arriver hand-offs in ba_point_trial and in the camera role,
speculative linearisation): vs_ba_solve over and over for SECONDS (default 30); every solve must reproduce the first one bit
for bit."""
import _env  # noqa: F401
import sys
import time

import numpy as np

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = Context(0)
w = ba_workload()
args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
ref = ctx.ba_solve(*args)
t0 = last = time.time()
n = 0
while time.time() - t0 < seconds:
    got = ctx.ba_solve(*args)
    assert np.array_equal(ref["poses"], got["poses"]) and np.array_equal(ref["points"], got["points"]), "solve %d differs" % n
    assert np.array_equal(ref["chi2_trace"], got["chi2_trace"]) and ref["trials"] == got["trials"]
    n += 1
    if time.time() - last > 10:
        last = time.time()
        print("%d solves identical so far" % n, flush=True)
print("soak ok: %d solves of cfg4 (10 x 2000, %d trials each), all identical to the first" % (n, ref["trials"]))
ctx.close()
