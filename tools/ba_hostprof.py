"""dev tool: host-side overhead of one motion-only BA call (max_iterations=0 -> setup + one launch + read-back)."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os, sys, time
import numpy as np
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
det, mat, ba = harness.gpu_callables(ctx)
xy0, d0 = det(frames[0])
lm = harness.LocalMapArrays(harness.backproject(xy0, depth0))
for k in range(1, 20):
    xy, d = det(frames[k]); mq, mt = mat(d0, d)
    lm.add_frame(lm.poses[-1], mq, xy[mt])
prob = lm.problem()
for it in (0, 1, 10):
    for _ in range(3): ctx.ba_solve(*prob, max_iterations=it)
    t0 = time.perf_counter()
    for _ in range(30): r = ctx.ba_solve(*prob, max_iterations=it)
    print("max_iterations %2d: %.1f us per call (iterations %d trials %d)" % (it, (time.perf_counter() - t0) / 30 * 1e6, r["iterations"], r["trials"]))
