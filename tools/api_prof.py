"""dev tool: cProfile of the class-API tracking period (harness.track_sequence_api) on the GPU: where the host time of the
unmodified main.py:181-214 call sequence goes."""
import _env  # noqa: F401
import cProfile
import pstats
import time

from visual_slam_amd import Context, harness

ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
for _ in range(3):
    harness.track_sequence_api(frames, depth0, context=ctx)
t0 = time.perf_counter()
for _ in range(10):
    harness.track_sequence_api(frames, depth0, context=ctx)
print("class API: %.1f us per frame" % ((time.perf_counter() - t0) / 10 / 20 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    harness.track_sequence_api(frames, depth0, context=ctx)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
ctx.close()
