import sys; sys.path.insert(0, "tools"); import _env
import time, numpy as np
from visual_slam_amd import Context, harness
from visual_slam_amd.harness import backproject
from visual_slam_amd.workloads import ICL_NUIM_K
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
for _ in range(3): harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)
acc = {}
N = 30
for _ in range(N):
    t = [time.perf_counter()]
    xy0, _, desc0 = ctx.detect_describe_bgr(frames[0], 20, 3000); t.append(time.perf_counter())
    X = backproject(xy0, depth0); t.append(time.perf_counter())
    ctx.track_begin(X, desc0, np.eye(4), ICL_NUIM_K, max_frames=19, pnp_iterations=100); t.append(time.perf_counter())
    calls = []
    for k in list(range(1, 20)) + [None]:
        t0 = time.perf_counter()
        ctx.track_frame_pipelined(frames[k] if k is not None else None, seed=k or 0, want_matches=False)
        calls.append(time.perf_counter() - t0)
    t.append(time.perf_counter())
    ctx.track_end(); t.append(time.perf_counter())
    for name, v in zip(("detect key frame", "backproject", "track_begin", "20 pipelined calls", "track_end"), np.diff(t)):
        acc[name] = acc.get(name, 0) + v
    for i, c in enumerate(calls):
        acc["call %02d" % i] = acc.get("call %02d" % i, 0) + c
tot = sum(v for k, v in acc.items() if not k.startswith("call"))
print("period %.1f us = %.1f us per frame" % (tot / N * 1e6, tot / N / 20 * 1e6))
for k, v in acc.items():
    print("  %-22s %8.1f us" % (k, v / N * 1e6))
ctx.close()
