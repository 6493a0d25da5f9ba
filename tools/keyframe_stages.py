"""dev tool: where the time of a key-frame insertion goes (slam.run_sequence's key-frame block = main.py:221-345), statement
by statement, without a profiler's per-call overhead -- the driver's `stages` hook calls lap(name) after every statement of the
block; the pieces inside localBundleAdjustement (graph from the structure-of-arrays mirror, vs_ba_solve, write-back) and inside
open_period are timed by wrappers.  Both driver modes: class API only, and the explicit device-resident tracking period.

    python tools/keyframe_stages.py [keyframe_gap=4] [repeats=20]        (on the GPU box; output -> profiles/rNN_keyframe_stages.txt)
"""
import _env  # noqa: F401
import sys
import time
from collections import OrderedDict

import numpy as np

from visual_slam_amd import Context, harness, slam
from visual_slam_amd import LocalBA as _lba
from visual_slam_amd import map as _map
from visual_slam_amd.workloads import ICL_NUIM_K

gap = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
be = slam.Backends(context=ctx)


class Laps:
    def __init__(self):
        self.acc = OrderedDict()
        self.inner = OrderedDict()
        self.t = time.perf_counter()
        self.keyframes = 0

    def lap(self, name):
        t = time.perf_counter()
        self.acc[name] = self.acc.get(name, 0.0) + (t - self.t)
        self.t = t


laps = [None]


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            if laps[0] is not None:
                laps[0].inner[label] = laps[0].inner.get(label, 0.0) + (time.perf_counter() - t0)
    setattr(obj, name, g)


timed(_lba.BundleAdjustment, "_graph_from_soa", "localBundleAdjustement: _graph_from_soa")
timed(_lba.BundleAdjustment, "optimize", "localBundleAdjustement: optimize (array conversion + vs_ba_solve)")
timed(ctx, "ba_solve", "localBundleAdjustement:   ctx.ba_solve (vs_ba_solve incl. upload / read-back)")
timed(ctx, "track_begin", "open_period:   ctx.track_begin")
timed(ctx, "track_end", "ctx.track_end")
timed(ctx, "triangulate_dlt", "triangulate:   ctx.triangulate_dlt")
timed(_map.Map, "soa", "Map.soa (all callers)")
timed(_map.Map, "_absorb_added", "Map._absorb_added (new points into the mirror)")
timed(_map.Map, "_flush", "Map._flush (pending observation batches into the Point objects)")


def run(resident, collect):
    L = Laps() if collect else None
    laps[0] = L
    if L:
        L.t = time.perf_counter()
    t0 = time.perf_counter()
    r = slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=gap, resident_ctx=ctx if resident else None, stages=L)
    dt = time.perf_counter() - t0
    laps[0] = None
    return r, dt, L


for resident in (False, True):
    for _ in range(3):
        run(resident, False)
    plain = [run(resident, False)[1] for _ in range(N)]
    acc, inner, total = OrderedDict(), OrderedDict(), []
    for _ in range(N):
        r, dt, L = run(resident, True)
        total.append(dt)
        for k, v in L.acc.items():
            acc[k] = acc.get(k, 0.0) + v
        for k, v in L.inner.items():
            inner[k] = inner.get(k, 0.0) + v
    nkf = len(r["keyframes"]) - 1
    med = float(np.median(plain))
    print("== driver, %s: %d frames, key frames at %s (gap %d), %d map points" % (
        "explicit device-resident tracking period" if resident else "class API only", len(frames), r["keyframes"], gap, r["n_points"]))
    print("   %.2f ms per run = %.0f frames/s (median of %d, uninstrumented); instrumented %.2f ms" % (
        med * 1e3, len(frames) / med, N, float(np.median(total)) * 1e3))
    kf_total = sum(v for k, v in acc.items() if not k.startswith("("))
    print("   key-frame block: %.1f us per key frame (%d key frames per run), %.1f %% of the run" % (
        kf_total / N / nkf * 1e6, nkf, 100 * kf_total / sum(total)))
    for k, v in acc.items():
        if k.startswith("("):
            print("   %-66s %8.1f us per tracked frame" % (k, v / N / (len(frames) - 1) * 1e6))
        else:
            print("   %-66s %8.1f us per key frame" % (k, v / N / nkf * 1e6))
    print("   -- inside (wrappers; per key frame)")
    for k, v in inner.items():
        print("   %-66s %8.1f us" % (k, v / N / nkf * 1e6))
ctx.close()
