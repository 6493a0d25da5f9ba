"""dev tool: the device-resident tracking period alone (for rocprofv3 / timing)."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os, sys, time
import numpy as np
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
harness.track_sequence_resident(ctx, frames[:4], depth0)
best = 1e9
for _ in range(5):
    p, dt, nm = harness.track_sequence_resident(ctx, frames, depth0)
    best = min(best, dt)
print("resident: %.1f us per frame (%.0f frames/s)" % (best / 20 * 1e6, 20 / best))
best = 1e9
harness.track_sequence_resident(ctx, frames[:4], depth0, pipelined=True)
for _ in range(5):
    p2, dt, nm = harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)
    best = min(best, dt)
print("resident, pipelined: %.1f us per frame (%.0f frames/s); identical poses: %s" % (best / 20 * 1e6, 20 / best, np.array_equal(p, p2)))
