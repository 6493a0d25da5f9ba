"""dev tool: the full driver (slam.run_sequence, key frame every 5th frame) with the decoded frames in pageable against pinned memory,
both modes, interleaved in one process; and with the collector's generations frozen.   python tools/driver_pinned_ab.py [rounds=4]"""
import _env  # noqa: F401
import gc
import statistics
import sys
import time

from visual_slam_amd import Context, harness, slam
from visual_slam_amd.workloads import ICL_NUIM_K

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
pinned = [ctx.pin(f) for f in frames]
be = slam.Backends(context=ctx)


def med(fr, resident, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        slam.run_sequence(fr, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx if resident else None)
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


for resident in (True, False):
    med(frames, resident, 3)
    med(pinned, resident, 3)
    for r in range(rounds):
        a, b = med(frames, resident), med(pinned, resident)
        print("%s: pageable frames %.2f ms = %.0f frames/s, pinned frames %.2f ms = %.0f frames/s" % (
            "resident period" if resident else "class API only", a * 1e3, 20 / a, b * 1e3, 20 / b), flush=True)
junk = [[i, str(i), (i, i)] for i in range(2000000)]  # a heap the size a long-lived process has
for resident in (True, False):
    a = med(pinned, resident)
    gc.collect()
    gc.freeze()
    b = med(pinned, resident)
    gc.unfreeze()
    print("%s, pinned, 2 M live objects on the heap: %.2f ms; generations frozen: %.2f ms" % ("resident period" if resident else "class API only", a * 1e3, b * 1e3), flush=True)
ctx.close()
