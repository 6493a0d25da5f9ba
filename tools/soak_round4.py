"""dev tool: soak of what round 4 added behind the C ABI, for SECONDS each (default 30):
  (1) the detector reading pinned frames where they lie (per-band flags, halo rows from the neighbours): six frames alternating
      through ONE pinned buffer, every result compared with the first result of that frame;
  (2) the large bundle adjustment with its structure built on the device, the ring-buffered banded Cholesky and the register
      back substitution: a sliding-window scene of 450 000 observations and the ragged 300-camera scene in turn, every solve
      compared bit for bit with the first solve of that scene (which the tests compare with the host-built structure)."""
import _env  # noqa: F401
import sys
import time

import numpy as np

from visual_slam_amd import Context, harness
from visual_slam_amd.workloads import ba_sliding_window_workload, synthetic_frame

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = Context(0)

frames, _ = harness.load_sequence(4)
frames = list(frames) + [synthetic_frame(640, 480, s) for s in (2, 7)]
buf = ctx.pin(np.zeros_like(frames[0]))
ref = []
for f in frames:
    buf[...] = f
    ref.append(tuple(a.copy() for a in ctx.detect_describe_bgr(buf, 20, 3000)))
t0 = last = time.time()
n = 0
while time.time() - t0 < seconds:
    k = (n * 5 + n // 7) % len(frames)
    buf[...] = frames[k]
    got = ctx.detect_describe_bgr(buf, 20, 3000)
    assert all(np.array_equal(a, b) for a, b in zip(got, ref[k])), "frame %d (call %d) differs" % (k, n)
    n += 1
    if time.time() - last > 10:
        last = time.time()
        print("detector: %d frames identical so far" % n, flush=True)
print("detector soak ok: %d frames through one pinned buffer (read where they lie), all identical to the first result of their frame" % n)

scenes = [ba_sliding_window_workload(40, 45000, 10, seed=5), ba_sliding_window_workload(100, 60000, 9, seed=6)]
args, refs = [], []
for w in scenes:
    a = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], ctx.pin(w["obs_pose"]), ctx.pin(w["obs_point"]), ctx.pin(w["obs_uv"]), w["K"])
    args.append(a)
    refs.append(ctx.ba_solve(*a, max_iterations=3))
    assert ctx.ba_structure_on_device()
t0 = last = time.time()
n = 0
while time.time() - t0 < seconds:
    k = n % len(scenes)
    got = ctx.ba_solve(*args[k], max_iterations=3)
    r = refs[k]
    assert ctx.ba_structure_on_device()
    assert np.array_equal(r["poses"], got["poses"]) and np.array_equal(r["points"], got["points"]) and np.array_equal(r["chi2_trace"], got["chi2_trace"]), \
        "solve %d (scene %d) differs" % (n, k)
    n += 1
    if time.time() - last > 10:
        last = time.time()
        print("bundle adjustment: %d solves identical so far" % n, flush=True)
print("BA soak ok: %d large solves (structure built on the device, %d and %d observations), all identical to the first of their scene"
      % (n, len(scenes[0]["obs_pose"]), len(scenes[1]["obs_pose"])))
ctx.close()
