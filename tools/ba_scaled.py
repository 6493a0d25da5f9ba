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

from visual_slam_amd.workloads import ICL_NUIM_K  # noqa: E402


def scene(n_cams, n_points, window, seed=3):
    """cameras on a straight 0.1 m-spaced track looking down +z, points 2.5-5.5 m ahead; point j is observed by `window`
    consecutive cameras starting at a random one (sliding-window visibility as key-frame BA has it)."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy = ICL_NUIM_K
    poses = np.tile(np.eye(4), (n_cams, 1, 1))
    poses[:, 0, 3] = 0.1 * np.arange(n_cams)
    start = r.integers(0, n_cams - window + 1, n_points)
    centre = 0.1 * (start + window / 2)
    pts = np.stack([centre + r.uniform(-1.0, 1.0, n_points), r.uniform(-1.2, 1.2, n_points), r.uniform(2.5, 5.5, n_points)], 1)
    cam = (start[:, None] + np.arange(window)[None, :]).astype(np.int32)            # [P, window]
    pt = np.repeat(np.arange(n_points, dtype=np.int32)[:, None], window, 1)
    pc = pts[pt.ravel()] - poses[cam.ravel(), :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)
    uv += r.normal(0, 0.5, uv.shape)
    bad = r.random(len(uv)) < 0.02
    uv[bad] += r.uniform(-50, 50, (int(bad.sum()), 2))
    poses0 = poses.copy()
    poses0[1:, :3, 3] += r.normal(0, 0.01, (n_cams - 1, 3))
    pts0 = pts + r.normal(0, 0.03, pts.shape)
    fixed = np.zeros(n_cams, np.uint8)
    fixed[0] = 1
    return dict(poses=poses0, pose_fixed=fixed, points=pts0, point_fixed=np.zeros(n_points, np.uint8),
                obs_pose=cam.ravel(), obs_point=pt.ravel(), obs_uv=uv, K=ICL_NUIM_K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", type=int, default=100)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--window", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    w = scene(a.cams, a.points, a.window)
    n_obs = len(w["obs_pose"])
    from visual_slam_amd.context import Context
    ctx = Context()
    kw = dict(huber_delta=np.sqrt(5.991), max_iterations=a.iters)
    if os.environ.get("PINNED", "1") == "1":  # the observation arrays in pinned memory: DMA-ed from where they lie
        for k in ("obs_pose", "obs_point", "obs_uv"):
            w[k] = ctx.pin(np.ascontiguousarray(w[k]))
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    g = ctx.ba_solve(*args, **kw)
    t0 = time.perf_counter()
    g = ctx.ba_solve(*args, **kw)
    dt = time.perf_counter() - t0
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
