"""Banded windows of several tiles: ba_schur_window (FP64 matrix cores, points ordered by lowest camera) against the tile
kernel (vs_tune_ba variant 3) and the CPU oracle on sliding-window scenes; median of five solves each; --big adds the scaled size.
  python tools/ba_window_check.py [--big]"""
import _env  # noqa: F401
import sys
import time

import numpy as np

from ba_scaled import scene
from visual_slam_amd.context import Context


def run(ctx, w, iters, variant):
    ctx.tune_ba(schur_variant=variant)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    kw = dict(huber_delta=np.sqrt(5.991), max_iterations=iters)
    g = ctx.ba_solve(*args, **kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        g = ctx.ba_solve(*args, **kw)
        ts.append((time.perf_counter() - t0) * 1e3)
    return g, float(np.median(ts))


def main():
    ctx = Context()
    from oracle import oracle
    for cams, pts, win, seed in ((24, 3000, 10, 1), (40, 20000, 10, 2), (33, 5000, 16, 3), (30, 4000, 17, 4), (12, 500, 5, 5), (100, 2000, 10, 6)):
        w = scene(cams, pts, win, seed)
        a, ta = run(ctx, w, 4, 0)
        b, tb = run(ctx, w, 4, 3)
        args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
        c = oracle.ba_solve(*args, huber_delta=np.sqrt(5.991), max_iterations=4)
        d_ab = max(np.abs(a["poses"] - b["poses"]).max(), np.abs(a["points"] - b["points"]).max())
        d_ac = max(np.abs(a["poses"] - c["poses"]).max(), np.abs(a["points"] - c["points"]).max())
        d_bc = max(np.abs(b["poses"] - c["poses"]).max(), np.abs(b["points"] - c["points"]).max())
        print("%3d cameras %6d points window %2d: window-kernel %.2f ms, tile-kernel %.2f ms; |window - tile| %.2e, |window - oracle| %.2e, |tile - oracle| %.2e; chi2 %.9g / %.9g / %.9g, trials %d/%d/%d"
              % (cams, pts, win, ta, tb, d_ab, d_ac, d_bc, a["chi2_final"], b["chi2_final"], c["chi2_final"], a["trials"], b["trials"], c["trials"]), flush=True)
    if "--big" in sys.argv:
        w = scene(100, 200000, 10)
        for k in ("obs_pose", "obs_point", "obs_uv"):
            w[k] = ctx.pin(np.ascontiguousarray(w[k]))
        a, ta = run(ctx, w, 3, 0)
        b, tb = run(ctx, w, 3, 3)
        print("scaled: window-kernel %.2f ms, tile-kernel %.2f ms; |window - tile| %.2e; chi2 %.9g / %.9g" % (ta, tb, max(np.abs(a["poses"] - b["poses"]).max(), np.abs(a["points"] - b["points"]).max()), a["chi2_final"], b["chi2_final"]))
    ctx.tune_ba(schur_variant=0)


if __name__ == "__main__":
    main()
