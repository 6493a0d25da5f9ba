#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.npz: small input/output vectors of the CPU oracle, cross-checked against the
independent NumPy twin before they are written (SURVEY.md 8c "golden fixtures to generate and commit").

These vectors pin the oracle (and, in the -m gpu tests, the HIP kernels) against accidental change.  They are NOT
reference outputs: the reference's native libraries (cv2, g2o) do not exist in this environment, so parity with them
is unpinned.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import np_twin  # noqa: E402
from conftest import icl_frame  # noqa: E402
from oracle import oracle  # noqa: E402
from visual_slam_amd.workloads import ba_workload, match_workload, synthetic_frame  # noqa: E402


def main():
    import re
    pat = np.array(re.findall(r"\{\s*(-?\d+),\s*(-?\d+),\s*(-?\d+),\s*(-?\d+)\}",
                              open(os.path.join(ROOT, "include", "vs_brief_pattern.h")).read()), dtype=np.int64)
    out = {}
    # (1)+(2) detection / description: a 64x64 synthetic tile and ICL-NUIM frame 0
    tile = synthetic_frame(64, 64, 1)
    xy, sc, desc = oracle.detect_describe_bgr(tile, 20, 3000)
    g = np_twin.gray_mean3(tile)
    txy, tsc = np_twin.fast9_detect(g, 20, 15, 3000)
    tdesc, _ = np_twin.brief256(g, txy, pat)
    assert np.array_equal(xy, txy) and np.array_equal(sc, tsc) and np.array_equal(desc, tdesc)
    out.update(tile_bgr=tile, tile_xy=xy, tile_score=sc, tile_desc=desc,
               tile_score_map=oracle.fast9_score_map(g, 20, 3))
    assert np.array_equal(out["tile_score_map"], np_twin.fast9_score_map(g, 20, 3))
    bgr = icl_frame(0)
    xy, sc, desc = oracle.detect_describe_bgr(bgr, 20, 3000)
    g = np_twin.gray_mean3(bgr)
    txy, tsc = np_twin.fast9_detect(g, 20, 15, 3000)
    tdesc, _ = np_twin.brief256(g, txy, pat)
    assert np.array_equal(xy, txy) and np.array_equal(sc, tsc) and np.array_equal(desc, tdesc)
    out.update(icl0_xy=xy, icl0_score=sc, icl0_desc=desc)
    xy100, sc100, _ = oracle.detect_describe_bgr(bgr, 20, 100)  # cap: strongest 100, ties by index
    assert np.array_equal(xy100, np_twin.fast9_detect(g, 20, 15, 100)[0])
    out.update(icl0_xy_cap100=xy100)
    # (3) 256 x 256 Hamming table with planted duplicates and low-entropy rows (forced ties)
    q, t = match_workload(256, 256, n_dup=16, seed=7)
    t[200:232, 8:] = 0
    q[100:120, 8:] = 0
    idx, dist = oracle.hamming_knn2(q, t)
    tidx, tdist = np_twin.hamming_knn2(q, t)
    assert np.array_equal(idx, tidx) and np.array_equal(dist, tdist)
    mq, mt, md = oracle.match_ratio(q, t, 0.8)
    out.update(ham_q=q, ham_t=t, ham_idx=idx, ham_dist=dist, ham_mq=mq, ham_mt=mt, ham_md=md)
    # (4) BA: 3 cameras x 20 points (checked against the dense twin) and cfg4 (10 x 2000) traces
    w = ba_workload(n_cams=3, n_points=20, seed=5)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    r = oracle.ba_solve(*args, max_iterations=6)
    _, _, ttrace = np_twin.ba_lm_dense(*args, float(np.sqrt(5.991)), 6)
    assert np.allclose(r["chi2_trace"], ttrace, rtol=1e-5)
    out.update(ba_small_poses=r["poses"], ba_small_points=r["points"], ba_small_chi2=r["chi2_trace"],
               ba_small_lambda=r["lambda_trace"])
    w = ba_workload()
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    r = oracle.ba_solve(*args, max_iterations=10)
    out.update(ba_cfg4_poses=r["poses"], ba_cfg4_chi2=r["chi2_trace"], ba_cfg4_lambda=r["lambda_trace"],
               ba_cfg4_chi2_initial=np.array([r["chi2_initial"]]), ba_cfg4_points_head=r["points"][:16])
    # (5) the rows around the path: PnP-RANSAC, essential RANSAC, pose recovery -- seeded synthetic scenes;
    # cross-checks against independent NumPy where one exists (SVD of E, DLT triangulation, projection of the inliers)
    from test_pnp import scene as pnp_scene
    from test_twoview import scene as tv_scene, true_E
    from visual_slam_amd.workloads import ICL_NUIM_K
    X, uv, T, bad = pnp_scene(300, 0.25, 0.4, 12)
    r = oracle.pnp_ransac(X, uv, ICL_NUIM_K, np.eye(4), seed=12)
    assert r["found"] and not set(r["inliers"].tolist()) & set(bad.tolist())
    fx, fy, cx, cy = ICL_NUIM_K
    Xc = (X[r["inliers"]] - r["pose"][:3, 3]) @ r["pose"][:3, :3]
    err = np.hypot(fx * Xc[:, 0] / Xc[:, 2] + cx - uv[r["inliers"], 0], fy * Xc[:, 1] / Xc[:, 2] + cy - uv[r["inliers"], 1])
    assert err.max() < 8.5 and np.linalg.norm(r["pose"][:3] - T[:3]) < 5e-3
    out.update(pnp_obj=X, pnp_img=uv, pnp_pose=r["pose"], pnp_inliers=r["inliers"])
    x1, x2, R, t, _, _ = tv_scene(400, 100, 0.5, 13)
    e = oracle.essential_ransac(x1, x2, 3.0 / 480, seed=13)
    sv = np.linalg.svd(e["E"])[1]
    assert e["found"] and abs(sv[0] - 1) < 1e-9 and abs(sv[1] - 1) < 1e-9 and sv[2] < 1e-9
    sel = e["mask"] == 1
    rp = oracle.recover_pose(e["E"], x1[sel], x2[sel])
    P0, P1 = np.eye(4)[:3], np.hstack([rp["R"], rp["t"][:, None]])
    from oracle import np_reference
    tri = np_reference.triangulate(P0, P1, np.c_[x1[sel], np.ones(sel.sum())], np.c_[x2[sel], np.ones(sel.sum())])
    tri = tri / np.linalg.norm(tri, axis=1, keepdims=True) * np.sign(tri[:, 3:])
    assert np.abs(tri - rp["X"]).max() < 1e-9 and np.abs(rp["R"] - R).max() < 0.05
    out.update(tv_x1=x1, tv_x2=x2, tv_E=e["E"], tv_mask=e["mask"], tv_R=rp["R"], tv_t=rp["t"], tv_pose_mask=rp["mask"],
               tv_X=rp["X"])
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote oracle_golden.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
