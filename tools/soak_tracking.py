"""dev tool: soak test of the resident tracking period (one-launch motion-only BA with its polled mailboxes): the 20-frame
ICL-NUIM period over and over for SECONDS (default 60), pipelined and frame by frame in turn; every run must reproduce the
first run's poses bit for bit, and no call may fail.  Prints a progress line every ~10 s."""
import _env  # noqa: F401
import sys
import time

import numpy as np

from visual_slam_amd import Context, harness

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
ref, _, _ = harness.track_sequence_resident(ctx, frames, depth0)
t0 = last = time.time()
runs = 0
while time.time() - t0 < seconds:
    got, _, _ = harness.track_sequence_resident(ctx, frames, depth0, pipelined=bool(runs & 1))
    assert np.array_equal(ref, got), "run %d differs" % runs
    runs += 1
    if time.time() - last > 10:
        last = time.time()
        print("%d periods (%d frames) identical so far" % (runs, 19 * runs), flush=True)
print("soak ok: %d periods, %d frames, all identical to the first run" % (runs, 19 * runs))
ctx.close()
