"""dev tool: randomised GPU-vs-oracle sweep of the RANSAC rows (PnP, essential matrix, pose recovery) and small BA scenes.
Reports how often the two sides disagree on a discrete outcome (inlier sets, masks, LM trial counts)."""
import _env  # noqa: F401  (sys.path + VS_DATASET_DIR)
import os, sys
import numpy as np
from test_pnp import scene as pnp_scene
from test_twoview import scene as tv_scene
from oracle import oracle
from visual_slam_amd import Context
from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload
ctx = Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = np.random.default_rng(123)
bad = {"pnp": 0, "ess": 0, "rec": 0, "ba": 0}
worst = {"pnp": 0.0, "ess": 0.0, "rec": 0.0, "ba": 0.0}
for k in range(N):
    n = int(r.integers(6, 800)); frac = float(r.uniform(0, 0.5)); noise = float(r.uniform(0, 1.5))
    X, uv, T, _ = pnp_scene(n, frac, noise, 1000 + k, float(r.uniform(0.001, 0.08)))
    g, c = ctx.pnp_ransac(X, uv, ICL_NUIM_K, np.eye(4), seed=k), oracle.pnp_ransac(X, uv, ICL_NUIM_K, np.eye(4), seed=k)
    if g["found"] != c["found"] or not np.array_equal(g["inliers"], c["inliers"]):
        bad["pnp"] += 1
    else:
        worst["pnp"] = max(worst["pnp"], float(np.abs(g["pose"] - c["pose"]).max()))
    n = int(r.integers(8, 1500)); out = int(r.uniform(0, 0.4) * n)
    x1, x2, R, t, _, _ = tv_scene(n, out, float(r.uniform(0, 1.0)), 2000 + k)
    g, c = ctx.essential_ransac(x1, x2, 3.0 / 480, seed=k), oracle.essential_ransac(x1, x2, 3.0 / 480, seed=k)
    if g["found"] != c["found"] or not np.array_equal(g["mask"], c["mask"]):
        bad["ess"] += 1
    elif c["found"]:
        worst["ess"] = max(worst["ess"], float(np.abs(g["E"] - c["E"]).max()))
        sel = c["mask"] == 1
        gr, cr = ctx.recover_pose(c["E"], x1[sel], x2[sel]), oracle.recover_pose(c["E"], x1[sel], x2[sel])
        if not np.array_equal(gr["mask"], cr["mask"]) or np.abs(gr["R"] - cr["R"]).max() > 1e-9:
            bad["rec"] += 1
        else:
            worst["rec"] = max(worst["rec"], float(np.abs(gr["X"] - cr["X"]).max()))
    if k % 4 == 0:
        w = ba_workload(n_cams=int(r.integers(2, 12)), n_points=int(r.integers(10, 300)), seed=3000 + k,
                        visibility=float(r.uniform(0.4, 1.0)))
        args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
        g, c = ctx.ba_solve(*args), oracle.ba_solve(*args)
        rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], c["poses"]))
        worst["ba"] = max(worst["ba"], float(rel))
        if rel > 1e-4:
            bad["ba"] += 1
print("cases", N, "discrete mismatches", bad, "worst continuous differences", worst)
