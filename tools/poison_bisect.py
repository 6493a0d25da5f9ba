"""dev tool: the class-API tracking period on a context whose device allocations are poisoned (vs_debug_poison_alloc); with
VS_POISON_SKIP=n the n-th allocation is left alone -- run for n = -1, 0, 1, ... to find the buffer something assumes to be zero."""
import _env  # noqa: F401
import sys

import numpy as np

from visual_slam_amd import Context, harness

byte = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x7F
ctx = Context(0)
ctx.debug_poison_alloc(byte)
frames, depth0 = harness.load_sequence(20)
try:
    steps = sys.argv[2] if len(sys.argv) > 2 else "arpx"
    if "a" in steps:
        harness.track_sequence(*harness.gpu_callables(ctx), frames, depth0, pnp=harness.gpu_pnp(ctx))
    ref, _, _ = harness.track_sequence_resident(ctx, frames, depth0)
    if "p" in steps:
        harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)
    api, _ = harness.track_sequence_api(frames, depth0, context=ctx)
    print("ok: class-API period ran, max pose difference %.2e" % float(np.abs(api - ref).max()))
except Exception as e:
    print("FAILED: %s" % e)
ctx.close()
