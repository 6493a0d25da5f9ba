"""dev tool: cProfile of slam.run_sequence on the 20 fixture frames (pinned), class API only and with the explicit resident period;
top entries by own time.   python tools/driver_profile.py [keyframe_gap=4] [top=45]"""
import _env  # noqa: F401
import cProfile
import io
import pstats
import sys
import time

from visual_slam_amd import Context, harness, slam
from visual_slam_amd.workloads import ICL_NUIM_K

gap = int(sys.argv[1]) if len(sys.argv) > 1 else 4
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
be = slam.Backends(context=ctx)
for resident in (False, True):
    kw = dict(keyframe_gap=gap, resident_ctx=ctx if resident else None)
    for _ in range(3):
        slam.run_sequence(frames, depth0, ICL_NUIM_K, be, **kw)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        slam.run_sequence(frames, depth0, ICL_NUIM_K, be, **kw)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("== %s: median %.2f ms per run (%.0f frames/s), pinned frames" % ("resident period" if resident else "class API only", ts[5] * 1e3, 20 / ts[5]))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        slam.run_sequence(frames, depth0, ICL_NUIM_K, be, **kw)
    pr.disable()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(top)
        txt = s.getvalue()
        print(txt[txt.index("ncalls") - 4:] if "ncalls" in txt else txt)
ctx.close()
