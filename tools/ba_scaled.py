"""one scaled BA run (100 cameras x 200 000 points, ~10 observations per point) where the linearisation
actually moves bytes.  Reports ms per LM trial, residuals/s and the effective HBM rate at 178 B/residual.
  python tools/ba_scaled.py [--cams 100] [--points 200000] [--window 10] [--iters 3] [--check]
--check also runs the CPU oracle on the same scene and compares chi2 traces and poses."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import argparse
import os
import sys
import time

import numpy as np

from visual_slam_amd.workloads import ICL_NUIM_K, ba_sliding_window_workload  # noqa: E402,F401


def scene(n_cams, n_points, window, seed=3):
    return ba_sliding_window_workload(n_cams, n_points, window, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=100)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--win-per", type=int, default=0, help="points per slab of ba_schur_window (vs_tune_ba; 0: the plan's own)")
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    w = scene(a.cams, a.points, a.window)
    n_obs = len(w["obs_pose"])
    from visual_slam_amd.context import Context
    ctx = Context()
    if a.win_per:
        ctx.tune_ba(points_per_workgroup=a.win_per)
    kw = dict(huber_delta=np.sqrt(5.991), max_iterations=a.iters)
    if os.environ.get("PINNED", "1") == "1":  # the observation arrays in pinned memory: DMA-ed from where they lie
        for k in ("obs_pose", "obs_point", "obs_uv"):
            w[k] = ctx.pin(np.ascontiguousarray(w[k]))
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    g = ctx.ba_solve(*args, **kw)
    dt = 1e9
    for _ in range(max(a.reps, 1)):
        t0 = time.perf_counter()
        g = ctx.ba_solve(*args, **kw)
        dt = min(dt, time.perf_counter() - t0)
    trials = max(int(g["trials"]), 1)
    print("scene: %d cameras, %d points, %d residuals; reduced system %d x %d" % (a.cams, a.points, n_obs, 6 * (a.cams - 1), 6 * (a.cams - 1)))
    print("GPU: %.1f ms per solve (%d iterations, %d trials) = %.2f ms/trial, %.1f Mresiduals/s, %.1f GB/s at 178 B/residual; chi2 %.6g -> %.6g"
          % (dt * 1e3, g["iterations"], trials, dt * 1e3 / trials, n_obs * trials / dt / 1e6, 178 * n_obs * trials / dt / 1e9,
             g["chi2_initial"], g["chi2_final"]))
    if a.check:
        from oracle import oracle
        t0 = time.perf_counter()
        c = oracle.ba_solve(*args, **kw)
        print("oracle: %.1f ms; chi2 %.6g -> %.6g; trials %d" % ((time.perf_counter() - t0) * 1e3, c["chi2_initial"], c["chi2_final"], c["trials"]))
        rel = max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(g["poses"], c["poses"]))
        print("max relative pose difference %.3g, max point difference %.3g" % (rel, np.abs(g["points"] - c["points"]).max()))


if __name__ == "__main__":
    main()
