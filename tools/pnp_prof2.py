"""dev tool: per-call times of Context.pnp_ransac over seeds (looks for outliers)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_pnp import scene
from visual_slam_amd.context import Context
from visual_slam_amd.workloads import ICL_NUIM_K
ctx = Context()
if os.environ.get("NOGC"):
    import gc
    gc.disable()
for n, frac, motion in ((420, 0.02, 0.003), (420, 0.3, 0.03), (3000, 0.3, 0.03)):
    X, uv, T, _ = scene(n, frac, 0.3, 1, motion)
    for _ in range(3):
        ctx.pnp_ransac(X, uv, ICL_NUIM_K, np.eye(4), seed=1)
    ts = []
    for k in range(50):
        t0 = time.perf_counter()
        r = ctx.pnp_ransac(X, uv, ICL_NUIM_K, np.eye(4), seed=k)
        ts.append((time.perf_counter() - t0) * 1e6)
    ts = np.array(ts)
    print("n=%d frac=%.2f: median %.1f us, max %.1f us at seed %d, >2x median: %s" % (n, frac, np.median(ts), ts.max(), ts.argmax(), np.where(ts > 2 * np.median(ts))[0].tolist()))
