"""dev tool: frames/s of the tracking period through the class API vs through the array harness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import Context, harness
ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
harness.track_sequence_api(frames[:3], depth0, context=ctx)
import cProfile, pstats
poses, dt = harness.track_sequence_api(frames, depth0, context=ctx)
print("class API: %.1f ms for 20 frames = %.1f frames/s" % (dt * 1e3, 20 / dt))
p2, st, _ = harness.track_sequence(*harness.gpu_callables(ctx), frames, depth0)
print("max pose diff api vs arrays:", max(np.linalg.norm(a - b) for a, b in zip(poses, p2)))
pr = cProfile.Profile(); pr.enable(); harness.track_sequence_api(frames, depth0, context=ctx); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(24)
