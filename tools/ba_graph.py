"""dev tool: BASELINE configs[3] (10 cameras x 2000 points, 10 LM iterations) with the batch of LM slots launched kernel by kernel
(the product) and replayed as one captured hipGraph (vs_tune_ba_graph); the interval compared is the batch alone -- first enqueue
(or graph launch) to results on the host -- capture and instantiation excluded.  Also the driver's key-frame problem sizes."""
import _env  # noqa: F401
import ctypes as C
import statistics

import numpy as np

from visual_slam_amd import Context
from visual_slam_amd.workloads import ba_workload

ctx = Context(0)
lib, h = ctx._lib, ctx.handle


def batch_us():
    v = C.c_double(0)
    lib.vs_tune_ba_graph(h, -1, C.addressof(v))
    return v.value


for label, kw in (("cfg4 10 x 2000", {}), ("4 cameras x 700 points (the driver's key-frame BA)", dict(n_cams=4, n_points=700, seed=5, visibility=0.6))):
    w = ba_workload(**kw)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    res = {}
    for mode in (2, 1, 2, 1):  # 2 = launch by launch, the uploads waited for first (the interval the graph form measures)
        lib.vs_tune_ba_graph(h, mode, None)
        for _ in range(5):
            g = ctx.ba_solve(*args)
        ts = []
        for _ in range(40):
            g = ctx.ba_solve(*args)
            ts.append(batch_us())
        res.setdefault(mode, []).append((statistics.median(ts), min(ts), g))
    a, b = res[2], res[1]
    same = all(np.array_equal(x[2]["poses"], a[0][2]["poses"]) and x[2]["trials"] == a[0][2]["trials"] for x in a + b)
    print("%s: %d LM trials per solve; batch of slots + export + read-back, wall us (median / min of 40, two rounds each):" % (label, a[0][2]["trials"]))
    print("   launch by launch : %.1f / %.1f   %.1f / %.1f" % (a[0][0], a[0][1], a[1][0], a[1][1]))
    print("   one hipGraph     : %.1f / %.1f   %.1f / %.1f   (capture + instantiation not included)" % (b[0][0], b[0][1], b[1][0], b[1][1]))
    print("   identical results: %s" % same)
lib.vs_tune_ba_graph(h, 0, None)
ctx.close()
