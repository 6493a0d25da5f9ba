import sys; sys.path.insert(0, "tools"); import _env
import time, numpy as np
from visual_slam_amd import Context, harness
from visual_slam_amd.workloads import synthetic_frame
ctx = Context(0)
frames, _ = harness.load_sequence(3)
for name, img in (("icl", ctx.pin(frames[0])), ("synthetic", ctx.pin(synthetic_frame()))):
    for _ in range(20): ctx.detect_describe_bgr(img, 20, 3000)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); ctx.detect_describe_bgr(img, 20, 3000); ts.append(time.perf_counter() - t0)
    print("%s: host ABI detect+describe median %.1f us, min %.1f" % (name, np.median(ts) * 1e6, min(ts) * 1e6))
ctx.close()
