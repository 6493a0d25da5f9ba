"""dev tool: where does a motion-only BA call spend its time (host prep vs kernel)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
det, mat, ba = harness.gpu_callables(ctx)
xy0, d0 = det(frames[0])
lm = harness.LocalMapArrays(harness.backproject(xy0, depth0))
for k in range(1, 20):
    xy, d = det(frames[k]); mq, mt = mat(d0, d)
    lm.add_frame(lm.poses[-1], mq, xy[mt])
    if k in (1, 5, 10, 19):
        prob = lm.problem()
        for _ in range(3): ba(*prob)
        t0 = time.perf_counter()
        for _ in range(20): r = ba(*prob)
        dt = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20): lm.problem()
        dp = (time.perf_counter() - t0) / 20
        print("frames %2d obs %5d : ba_solve %.1f us (problem() %.1f us) iterations %d trials %d" % (k, len(prob[4]), dt * 1e6, dp * 1e6, r["iterations"], r["trials"]))
    res = ba(*lm.problem())
    for i in range(1, len(lm.poses)): lm.poses[i] = res["poses"][i]
for n in (1, 3):
    t0 = time.perf_counter()
    for _ in range(50): det(frames[n])
    print("detect_describe host call %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
xy, d = det(frames[1])
t0 = time.perf_counter()
for _ in range(50): mat(d0, d)
print("match_ratio host call %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
