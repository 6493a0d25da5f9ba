"""dev tool: per-step phase stamps of ba_motion_persistent (vs_mo_profile) inside the resident tracking period, frame by frame."""
import _env  # noqa: F401
import sys

import numpy as np

from visual_slam_amd import Context
from visual_slam_amd.harness import backproject, load_sequence
from visual_slam_amd.workloads import ICL_NUIM_K

ctx = Context(0)
lib = ctx._lib
frames, depth0 = load_sequence(20)
frames = [ctx.pin(f) for f in frames]
names = ["rendezvous", "decision", "linearise+reduce", "6x6 solve+record", "trial chi2", "post"]
lib.vs_mo_profile(ctx.handle, 1)
acc = {}
for rep in range(3):
    xy0, _, desc0 = ctx.detect_describe_bgr(frames[0], 20, 3000)
    ctx.track_begin(backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=19, pnp_iterations=100)
    for k in range(1, 20):
        ctx.track_frame(frames[k], seed=k, want_matches=False)
        out = np.zeros((64, 8))
        n = lib.vs_mo_profile_read(ctx.handle, out.ctypes.data, 64)
        if rep == 2 and n > 1:
            r = out[:n]
            wall = r[n - 1, 7] / max(n - 1, 1)
            cyc_per_us = (r[n - 1, 0] - r[0, 0]) / max(r[n - 1, 7], 1e-9)
            ph = {nm: [] for nm in names}
            for s in range(1, n - 1):   # full steps (a step that found the solve finished has no phases)
                t = r[s]
                seq = [t[0], t[1], t[2], t[3] if t[3] else t[2], t[4], t[5], t[6]]
                if not t[4]:
                    continue
                for nm, a, b in zip(names, seq[:-1], seq[1:]):
                    ph[nm].append((b - a) / cyc_per_us)
            acc[k] = (n, wall, cyc_per_us, {nm: (np.mean(v) if v else 0.0) for nm, v in ph.items()}, sum(1 for s in range(1, n - 1) if r[s, 3]))
    ctx.track_end()
lib.vs_mo_profile(ctx.handle, 0)
print("frame cameras steps  us/step  clock(MHz)  lin-steps | mean microseconds per phase of a full step")
for k, (n, wall, cpu, ph, nlin) in sorted(acc.items()):
    print("%5d %7d %5d %8.2f %10.0f %9d | %s" % (k, k, n, wall, cpu, nlin, "  ".join("%s %.2f" % (nm, ph[nm]) for nm in names)))
ctx.close()
