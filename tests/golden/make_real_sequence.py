#!/usr/bin/env python3
"""Writes tests/golden/real_sequence.json + real_sequence.npz + real_ba_{early,middle,last}.npz: the headless driver
(visual_slam_amd/slam.py = the control flow of the reference's src/v2/main.py:150-348) run with the CPU-ORACLE back ends over the
first N (default 420) frames of the reference's own data set, /root/reference/data/ICL_NUIM (living room, trajectory 3), with the
reference's own key-frame rule (main.py:221: more than 20 frames since the last key frame or fewer than 80 tracked points, and fewer
than 90 % of the last key frame's points tracked) and both initialisations (depth of frame 0; main.py:78-148 two-view).

Run in the build container only (it reads /root/reference/data; the GPU box has no reference):
    python tests/golden/make_real_sequence.py [N]

What is stored (data only):
  real_sequence.json   per run: absolute trajectory error against traj3.gt.freiburg (Sim(3)-aligned camera centres), key frames, tracked
                       points, map size, per key-frame interval the ratio estimated / true step length, and the sizes of every local
                       bundle adjustment the driver ran
  real_sequence.npz    the estimated trajectories [N,4,4] and the ground truth used
  real_ba_*.npz        the bundle-adjustment problems ACTUALLY handed to the solver at three key frames of the depth-initialised run
                       (early / middle / last: poses, fixed flags, points, observations, scale edges, Huber width) and what the oracle
                       made of them (chi2, trials) -- tests/test_real_sequence.py solves them on the HIP path against the oracle

ICL-NUIM note.  The data set's ground truth is written for the camera model of its renderer, fy = -480 (Handa et al. 2014: the
image y axis points the other way); this project -- like the reference, main.py:55-56 / src/v1/slam_test.py:144-145 -- uses
fy = +480, so its trajectories are the mirror image (y -> -y) of the published ones.  The ground truth is mirrored accordingly
(S G S, S = diag(1, -1, 1, 1)) before the alignment, which cannot undo a reflection.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DATA = os.environ.get("VS_REFERENCE_DATA", "/root/reference/data/ICL_NUIM")


def mirrored(gt):
    S = np.diag([1.0, -1.0, 1.0, 1.0])
    return np.array([S @ g @ S for g in gt])


def step_ratios(P, G, kf):
    out = []
    for a, b in zip(kf[:-1], kf[1:]):
        de, dg = np.linalg.norm(P[b][:3, 3] - P[a][:3, 3]), np.linalg.norm(G[b][:3, 3] - G[a][:3, 3])
        out.append(float(de / dg) if dg > 0 else 0.0)
    return out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 420
    from oracle import oracle
    from test_slam_driver import oracle_backends
    from visual_slam_amd import dataset, slam
    from visual_slam_amd.workloads import ICL_NUIM_K
    oracle.build()
    oracle.load()
    seq = dataset.Sequence(DATA, groundtruth="traj3.gt.freiburg")
    frames = [seq.rgb(i) for i in range(N)]
    depth0 = seq.depth(0)
    gt = mirrored(seq.gt[1][:N])
    summary = {"frames": N, "dataset": "ICL-NUIM living room traj3 (the reference's data/ICL_NUIM), frames 0..%d" % (N - 1),
               "back_ends": "CPU oracle (oracle/vs_oracle.c) through visual_slam_amd/slam.py", "keyframe_gap": 20, "min_tracked": 80,
               "ground_truth": "traj3.gt.freiburg, mirrored y -> -y (see the header of tests/golden/make_real_sequence.py)",
               "runs": {}}
    arrays = {"gt": gt}
    problems = []
    guarded_problems = []
    # two runs as main.py has it, and the same two with the driver's guards (NOT in the reference: parallax >= 1 degree and
    # reprojection error <= 2 px for a new point, key-frame poses adjusted with the points held fixed; slam.run_sequence)
    for init, guards in (("depth", False), ("two_view", False), ("depth", True), ("two_view", True)):
        name = init + ("_guarded" if guards else "")
        gkw = dict(new_point_min_parallax_deg=1.0, new_point_max_reproj_px=2.0, keyframe_ba="poses_only") if guards else {}
        be = oracle_backends(oracle)
        inner = be._solver
        calls = []

        def recording(*a, _inner=inner, _calls=calls, **k):
            r = _inner(*a, **k)
            # a local bundle adjustment: a free point, or (guarded runs) scale edges -- motion-only solves have neither
            if len(a[2]) and (not np.all(a[3]) or k.get("scale_edges")):
                _calls.append((a, k, r))
            return r
        be._solver = recording
        t0 = time.perf_counter()
        r = slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=20, min_tracked=80, init=init, **gkw)
        dt = time.perf_counter() - t0
        P, kf = r["poses"], r["keyframes"]
        ate = dataset.ate_rmse(P, gt)
        half = dataset.ate_rmse(P[:N // 2], gt[:N // 2])
        G0 = np.array([np.linalg.inv(gt[0]) @ g for g in gt])
        n350 = min(N, 350)
        first350 = dataset.ate_rmse(P[:n350], gt[:n350])
        summary["runs"][name] = {
            "guards": gkw if guards else None, "ate_rmse_first_350_frames_m": first350["rmse"], "gt_path_length_first_350_frames_m": first350["path_length"],
            "ate_rmse_m": ate["rmse"], "ate_mean_m": ate["mean"], "ate_max_m": ate["max"], "sim3_scale": ate["scale"],
            "gt_path_length_m": ate["path_length"], "ate_rmse_first_half_m": half["rmse"], "gt_path_length_first_half_m": half["path_length"],
            "keyframes": [int(k) for k in kf], "map_points": int(r["n_points"]),
            "tracked_min": int(min(r["tracked"])), "tracked_median": float(np.median(r["tracked"])),
            "pnp_inliers_min": int(min(r["pnp_inliers"])),
            "step_ratio_estimated_over_true_per_keyframe_interval": step_ratios(P, G0, kf),
            "local_ba": [{"poses": int(len(a[0])), "points": int(len(a[2])), "observations": int(len(a[4])),
                          "scale_edges": 0 if not k.get("scale_edges") else int(len(k["scale_edges"][0])),
                          "chi2_initial": float(res["chi2_initial"]), "chi2_final": float(res["chi2_final"]), "trials": int(res["trials"])}
                         for a, k, res in calls],
            "cpu_oracle_seconds": dt, "cpu_oracle_frames_per_s": N / dt}
        arrays["poses_" + name] = P
        # how far the map is from the data set's own depth images at every key frame: for the points a key frame observes, depth in
        # that camera (map units) / true depth at the observed pixel (metres) -- a constant ratio is a consistent scale; points first
        # seen at this key frame or the one before (the newly triangulated ones) listed apart from the older ones
        m = r["map"]
        diag = []
        for j, img in enumerate(kf):
            f = m.GetFrame(j)
            w2c = np.linalg.inv(np.asarray(f.GetPose(), np.float64))
            D = seq.depth(img)
            old, new = [], []
            for p in m.points_3d.values():
                o = p.frames.get(j)
                if o is None:
                    continue
                zt = D[int(o[1][1]), int(o[1][0])]
                if zt <= 0:
                    continue
                z = (w2c[:3, :3] @ np.asarray(p.location_3d, np.float64) + w2c[:3, 3])[2]
                (new if j >= 2 and min(p.frames.keys()) >= j - 1 else old).append(z / zt)
            q = lambda a: None if not a else [len(a), float(np.median(a)), float(np.percentile(a, 25)), float(np.percentile(a, 75))]  # noqa: E731
            diag.append({"keyframe": j, "image": int(img), "older_points_n_median_q25_q75": q(old), "new_points_n_median_q25_q75": q(new)})
        summary["runs"][name]["map_depth_over_true_depth_at_keyframes"] = diag
        if name == "depth":
            problems = calls
        if name == "depth_guarded":
            guarded_problems = calls
        print("%s: %.1f s, %d key frames, %d map points, ATE rmse %.3f m over %.2f m (first half: %.3f over %.2f)" % (
            name, dt, len(kf), r["n_points"], ate["rmse"], ate["path_length"], half["rmse"], half["path_length"]))
    # ---- three real problems: early (5 poses), middle, last
    picks = {"early": (problems, next(i for i, c in enumerate(problems) if len(c[0][0]) >= 5)), "middle": (problems, len(problems) // 2),
             "last": (problems, len(problems) - 1),
             # the guarded run's key-frame adjustment at a key frame with 21 free cameras: every point fixed, a scale edge per key frame
             "guarded": (guarded_problems, next(i for i, c in enumerate(guarded_problems) if len(c[0][0]) >= 22))}
    summary["ba_fixtures"] = {}
    for name, (plist, i) in picks.items():
        a, k, res = plist[i]
        se = k.get("scale_edges")
        out = dict(poses=np.asarray(a[0], np.float64), pose_fixed=np.asarray(a[1], np.uint8), points=np.asarray(a[2], np.float64),
                   point_fixed=np.asarray(a[3], np.uint8), obs_pose=np.asarray(a[4], np.int32), obs_point=np.asarray(a[5], np.int32),
                   obs_uv=np.asarray(a[6], np.float64), K=np.asarray(a[7], np.float64), huber_delta=np.float64(k.get("huber_delta", 0.0)),
                   max_iterations=np.int32(k.get("max_iterations", 10)), dcs_phi=np.float64(k.get("dcs_phi", 1.0)),
                   scale_parent=np.asarray(se[0] if se else [], np.int32), scale_child=np.asarray(se[1] if se else [], np.int32),
                   scale_meas=np.asarray(se[2] if se else [], np.float64),
                   oracle_chi2_initial=np.float64(res["chi2_initial"]), oracle_chi2_final=np.float64(res["chi2_final"]),
                   oracle_trials=np.int32(res["trials"]), oracle_poses=np.asarray(res["poses"], np.float64))
        assert k.get("obs_info") is None
        path = os.path.join(HERE, "real_ba_%s.npz" % name)
        np.savez_compressed(path, **out)
        obs_per_point = np.bincount(out["obs_point"], minlength=len(out["points"]))
        span = [int(out["obs_pose"][out["obs_point"] == p].max() - out["obs_pose"][out["obs_point"] == p].min()) for p in np.nonzero(obs_per_point)[0]]
        summary["ba_fixtures"][name] = {"local_ba_index": int(i), "poses": int(len(out["poses"])), "points": int(len(out["points"])),
                                        "observations": int(len(out["obs_pose"])), "scale_edges": int(len(out["scale_parent"])),
                                        "widest_camera_span_of_a_point": int(max(span)), "bytes": os.path.getsize(path)}
        print("real_ba_%s.npz: key frame %d: %d poses, %d points, %d observations, widest span %d, %d bytes" % (
            name, i + 1, len(out["poses"]), len(out["points"]), len(out["obs_pose"]), max(span), os.path.getsize(path)))
    np.savez_compressed(os.path.join(HERE, "real_sequence.npz"), **arrays)
    with open(os.path.join(HERE, "real_sequence.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote real_sequence.json / real_sequence.npz")


if __name__ == "__main__":
    main()
