import sys; sys.path.insert(0, "tools"); import _env
import time, numpy as np
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
for i in range(3):
    t0 = time.perf_counter()
    try:
        p, dt, nm = harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)
        print("run %d ok: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    except Exception as e:
        print("run %d FAILED after %.1f ms: %s" % (i, (time.perf_counter() - t0) * 1e3, e), flush=True)
        try:
            ctx.track_end()
        except Exception as e2:
            print("track_end:", e2)
ctx.close()
