"""dev tool: phase stamps of pnp_ransac_kernel (vs_pnp_profile) inside the resident tracking period, frame by frame."""
import _env  # noqa: F401
import ctypes as C
import statistics
import time

import numpy as np

from visual_slam_amd import Context
from visual_slam_amd.harness import backproject, load_sequence
from visual_slam_amd.workloads import ICL_NUIM_K

ctx = Context(0)
lib = ctx._lib
frames, depth0 = load_sequence(20)
frames = [ctx.pin(f) for f in frames]
H = 100
for profile in (0, 1):
    lib.vs_pnp_profile(ctx.handle, profile)
    for rep in range(3):
        xy0, _, desc0 = ctx.detect_describe_bgr(frames[0], 20, 3000)
        ctx.track_begin(backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=19, pnp_iterations=H)
        rows = []
        t0 = time.perf_counter()
        for k in range(1, 20):
            ctx.track_frame(frames[k], seed=k, want_matches=False)
            if profile and rep == 2:
                out = np.zeros((H + 1, 8))
                n = lib.vs_pnp_profile_read(ctx.handle, out.ctypes.data, H + 1)
                if n == H + 1:
                    rows.append(out.copy())
        dt = time.perf_counter() - t0
        ctx.track_end()
    print("profile=%d: %.1f us per frame (frame by frame)" % (profile, dt / 19 * 1e6))
for k, r in enumerate(rows):
    hyp, fin = r[:H], r[H]
    print('   after LM (medians): record %.1f, stored %.1f, scored %.1f, published %.1f' % tuple(statistics.median(hyp[:, c] - hyp[:, 2]) for c in (4, 5, 6, 3)))
    lm = hyp[:, 2] - hyp[:, 1]
    print("frame %2d  hyp: start %.1f..%.1f  sample %.1f  LM med %.1f max %.1f  published: first10 max %.1f, all max %.1f | "
          "fin: start %.1f  winner %.1f  listed %.1f  LM done %.1f  end %.1f"
          % (k + 1, hyp[:, 0].min(), hyp[:, 0].max(), statistics.median(hyp[:, 1] - hyp[:, 0]), statistics.median(lm), lm.max(),
             hyp[:10, 3].max(), hyp[:, 3].max(), fin[0], fin[1], fin[2], fin[4], fin[3]))
ctx.close()
