"""dev tool: timing of the general (Schur) BA path on BASELINE cfg4 (10 cameras x 2000 points) and a larger window.
The clocks of an idle GPU take a while to come up for latency-bound work: 150 solves are run, the median of the last 50 counts."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os
import statistics
import time

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

ctx = Context(0)
if os.environ.get("SCHUR"):  # "variant,points per workgroup,max slabs" -> the library's tuning hook
    ctx.tune_ba(*[int(v) for v in os.environ["SCHUR"].split(",")])
for (nc, npts, vis) in [(10, 2000, 1.0), (15, 5000, 0.7)]:
    w = ba_workload(n_cams=nc, n_points=npts, visibility=vis)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    ts = []
    for _ in range(150):
        t0 = time.perf_counter()
        r = ctx.ba_solve(*args)
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts[-50:])
    print("%d cams x %d pts (%d obs): ba_solve %.1f us (first 10: %.1f us), %d iterations %d trials -> %.1f us/trial"
          % (nc, npts, len(w["obs_pose"]), dt * 1e6, statistics.median(ts[:10]) * 1e6, r["iterations"], r["trials"],
             dt * 1e6 / max(r["trials"], 1)))
ctx.close()
