"""dev tool: cProfile of the headless driver (resident tracking between key frames)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import Context, harness, slam
from visual_slam_amd.workloads import ICL_NUIM_K
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
be = slam.Backends(context=ctx)
for _ in range(2):
    slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx)
t0 = time.perf_counter(); slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx); dt = time.perf_counter() - t0
print("resident driver: %.1f ms for 20 frames" % (dt * 1e3))
pr = cProfile.Profile(); pr.enable(); slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=4, resident_ctx=ctx); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
