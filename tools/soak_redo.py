"""dev tool: soak of the chained tracking period's redo path (round 4): the 20-frame ICL-NUIM period, pipelined, over and over
for SECONDS (default 40) with a hand-off fault injected into a different frame of every period (vs_track_debug: that frame's PnP
launch waits for a tag nobody publishes; every workgroup's bounded wait runs out after ~50 ms and the host redoes the frame
host-paced).  Every period must reproduce the undisturbed frame-by-frame poses bit for bit and count exactly one redo."""
import _env  # noqa: F401
import ctypes as C
import sys
import time

import numpy as np

from visual_slam_amd import Context, harness
from visual_slam_amd.harness import backproject
from visual_slam_amd.workloads import ICL_NUIM_K

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ctx = Context(0)
lib = ctx._lib
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
ref, _, _ = harness.track_sequence_resident(ctx, frames, depth0)


def redos():
    n = C.c_int(0)
    assert lib.vs_track_debug(ctx.handle, 0, C.byref(n)) == 0
    return n.value


t0 = last = time.time()
periods = 0
while time.time() - t0 < seconds:
    fail_at = 1 + periods % 19
    xy0, _, desc0 = ctx.detect_describe_bgr(frames[0], 20, 3000)
    ctx.track_begin(backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=19, pnp_iterations=100)
    before = redos()
    r = None
    for k in list(range(1, 20)) + [None]:
        if k == fail_at:
            lib.vs_track_debug(ctx.handle, 1, None)
        out = ctx.track_frame_pipelined(frames[k] if k is not None else None, seed=k or 0, want_matches=False)
        if out is not None:
            r = out
    ctx.track_end()
    assert redos() == before + 1, "period %d: %d redos" % (periods, redos() - before)
    assert np.array_equal(ref, r["poses"]), "period %d (fault in frame %d) differs" % (periods, fail_at)
    periods += 1
    if time.time() - last > 10:
        last = time.time()
        print("%d periods, one redo each, all identical so far" % periods, flush=True)
clean, _, _ = harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)
assert np.array_equal(ref, clean)
print("soak ok: %d periods (%d frames) with a hand-off fault injected into frames 1..19 in turn: %d redos, every period bit-identical to "
      "the frame-by-frame run; a clean pipelined period afterwards identical too" % (periods, 19 * periods, redos()))
ctx.close()
