"""Headless tracking harness: the call pattern of the reference's tracking loop (src/v2/main.py:173-214) -- detect+describe
-> match against the last key frame's map points -> PnP-RANSAC from the previous pose -> motion-only BA over the local
map.  (Key-frame insertion with triangulation + local BA lives in slam.py.)

What is NOT reproduced: the two-view essential-matrix initialisation.  The harness initialises the
map from frame 0's keypoints back-projected with the dataset's depth image (camera 0 = world); no ground-truth pose is
used.

Data: callers name the sequence directory (data_dir / $VS_DATASET_DIR); tests and bench.py use the first 20 RGB frames of
ICL-NUIM living-room trajectory 3 (the trajectory present in the reference's data/; BASELINE.json names traj0, which is
not there -- SURVEY.md 0) that the repository holds as fixtures.
"""
import os
import time

import numpy as np

from .frame import imread
from .workloads import ICL_NUIM_K

HUBER = float(np.sqrt(5.991))


def dataset_dir(data_dir=None):
    """Directory of an ICL-NUIM / TUM style sequence (rgb/N.png, depth/N.png): the argument, else $VS_DATASET_DIR.  The
    package ships no data; tests and bench.py point this at the fixture frames they hold."""
    d = data_dir or os.environ.get("VS_DATASET_DIR")
    if not d:
        raise ValueError("no dataset directory: pass data_dir or set VS_DATASET_DIR")
    return d


def load_sequence(n_frames=20, data_dir=None):
    d = dataset_dir(data_dir)
    frames = [imread(os.path.join(d, "rgb", "%d.png" % i)) for i in range(n_frames)]
    if any(f is None for f in frames):
        raise FileNotFoundError("sequence frames missing under %s" % d)
    from PIL import Image
    depth0 = np.asarray(Image.open(os.path.join(d, "depth", "0.png"))).astype(np.float64) / 5000.0  # metres
    return frames, depth0


def backproject(xy, depth, K=ICL_NUIM_K):
    fx, fy, cx, cy = K
    z = depth[xy[:, 1].astype(int), xy[:, 0].astype(int)]
    return np.stack([(xy[:, 0] - cx) * z / fx, (xy[:, 1] - cy) * z / fy, z], 1)


class LocalMapArrays:
    """SoA local map of one tracking period: key frame 0 fixed, map points fixed, one free pose per tracked frame and
    the accumulated observations -- what motionOnlyBundleAdjustement builds from the object graph every frame
    (LocalBA.py:195-214), kept as arrays so nothing is rebuilt."""

    def __init__(self, points):
        self.points = np.ascontiguousarray(points, np.float64)
        self.poses = [np.eye(4)]
        self.obs_pose, self.obs_point, self.obs_uv = [], [], []

    def add_frame(self, pose, point_idx, uv):
        k = len(self.poses)
        self.poses.append(np.array(pose, np.float64))
        self.obs_pose.append(np.full(len(point_idx), k, np.int32))
        self.obs_point.append(np.asarray(point_idx, np.int32))
        self.obs_uv.append(np.asarray(uv, np.float64))

    def problem(self):
        n = len(self.poses)
        fixed = np.zeros(n, np.uint8)
        fixed[0] = 1
        return (np.stack(self.poses), fixed, self.points, np.ones(len(self.points), np.uint8),
                np.concatenate(self.obs_pose), np.concatenate(self.obs_point), np.concatenate(self.obs_uv), ICL_NUIM_K)


def track_sequence(detect, match, ba, frames, depth0, max_kp=3000, pnp=None):
    """One tracking period.  detect(bgr) -> (xy, desc); match(q, t) -> (mq, mt); pnp(obj, img, K4, pose0, seed=) ->
    dict(found, pose, inliers) or None to start BA from the previous pose; ba(*problem) -> dict(poses=...).
    Returns (poses [n,4,4], stage seconds dict, per-frame match counts)."""
    t_det = t_match = t_pnp = t_ba = 0.0
    t0 = time.perf_counter()
    xy0, desc0 = detect(frames[0])
    t_det += time.perf_counter() - t0
    lm = LocalMapArrays(backproject(xy0, depth0))
    n_matches = []
    for k in range(1, len(frames)):
        t0 = time.perf_counter()
        xy, desc = detect(frames[k])
        t1 = time.perf_counter()
        mq, mt = match(desc0, desc)
        t2 = time.perf_counter()
        pose = lm.poses[-1]
        if pnp is not None and len(mq) >= 5:  # main.py:191-204: previous pose as the extrinsic guess
            r = pnp(lm.points[mq], xy[mt], ICL_NUIM_K, pose, seed=k)
            if r["found"]:
                pose = r["pose"]
        t3 = time.perf_counter()
        lm.add_frame(pose, mq, xy[mt])
        res = ba(*lm.problem())
        for i in range(1, len(lm.poses)):
            lm.poses[i] = res["poses"][i]
        t4 = time.perf_counter()
        t_det += t1 - t0
        t_match += t2 - t1
        t_pnp += t3 - t2
        t_ba += t4 - t3
        n_matches.append(len(mq))
    stages = {"detect_describe": t_det, "match": t_match, "pnp_ransac": t_pnp, "motion_ba": t_ba}
    return np.stack(lm.poses), stages, n_matches


def track_sequence_resident(ctx, frames, depth0, pnp=True, pipelined=False):
    """The same tracking period through the device-resident session (vs_track_begin / vs_track_frame): the key frame's
    map is uploaded once, every frame uploads only its image.  pipelined=True (recorded streams): frame k+1 is uploaded,
    detected and matched on a second stream while frame k's PnP + BA run.  Returns (poses [n,4,4], seconds, per-frame
    match counts)."""
    t0 = time.perf_counter()
    xy0, _, desc0 = ctx.detect_describe_bgr(frames[0], 20, 3000)
    ctx.track_begin(backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=max(len(frames) - 1, 1),
                    pnp_iterations=100 if pnp else 0)
    n_matches, r = [], None
    if pipelined:
        for k in list(range(1, len(frames))) + [None]:
            out = ctx.track_frame_pipelined(frames[k] if k is not None else None, seed=k or 0, want_matches=False)
            if out is not None:
                r = out
                n_matches.append(r["n_matches"])
    else:
        for k in range(1, len(frames)):
            r = ctx.track_frame(frames[k], seed=k, want_matches=False)
            n_matches.append(r["n_matches"])
    ctx.track_end()
    dt = time.perf_counter() - t0
    poses = r["poses"] if r is not None else np.eye(4)[None]
    return poses, dt, n_matches


def track_sequence_api(frames, depth0, context=None, ba_solver=None, pnp=True, pnp_solver=None, verbatim=False):
    """The same tracking period written against the reference's class API as src/v2/main.py:173-214 uses it:
    Frame.process_frame -> Map.GetImagePointsWithFrameID -> FeatureMatcher.match_features -> solvePnPRansac ->
    Map.AddParentAndPose / AddPointToFrameCorrespondences -> BundleAdjustment.motionOnlyBundleAdjustement.
    verbatim=False: the PnP call is handed float64 object points and the previous pose as its (world-to-camera) guess, per frame
    seed k -- the period is then the array path's, pose for pose.  verbatim=True: the statements of main.py:187-204 as they
    stand -- the object points as .astype(np.float32), rvec / tvec guesses taken from W_T_prev itself (camera-to-world where
    OpenCV expects world-to-camera), no check of retval, the default seed.  Returns (poses [n,4,4], seconds)."""
    from .LocalBA import BundleAdjustment, Camera, Isometry3d
    from .frame import FeatureExtractor, FeatureMatcher, Frame
    from .map import Map
    from .point import Point
    from . import helper_functions as hf
    extractor, matcher = FeatureExtractor(context=context), FeatureMatcher(context=context)
    camera = Camera(*ICL_NUIM_K)
    fx, fy, cx, cy = ICL_NUIM_K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    t0 = time.perf_counter()
    key = Frame(frames[0], None, 0)
    key.AddPose(np.eye(4))
    key.SetAsKeyFrame()
    kp0, ft0, _ = key.process_frame(extractor)
    local_map = Map()
    local_map.AddFrame(0, key)
    for i, (X, uv, d) in enumerate(zip(backproject(kp0, depth0), kp0, ft0)):
        pt = Point(location=X, id=i + 1)
        pt.AddFrame(frame=key, uv=uv, descriptor=d)
        local_map.AddPoint3D(point_id=i + 1, point_3d=pt)
    for k in range(1, len(frames)):
        cur = Frame(frames[k], None, k)
        kp_cur, ft_cur, _ = cur.process_frame(extractor)
        kp_prev, ft_prev, known_3d, point_ids = local_map.GetImagePointsWithFrameID(0)
        matches, _, _, cur_pts, cur_fts = matcher.match_features(kp_prev, ft_prev, kp_cur, ft_cur)
        prev_pose = np.asarray(local_map.GetFrame(k - 1).GetPose(), np.float64)
        if verbatim:  # main.py:187-204
            known_3d_matched = np.array([known_3d[m[0].queryIdx] for m in matches])
            W_T_prev = local_map.GetFrame(k - 1).GetPose()
            rvec_guess = hf.Rtorvec(W_T_prev[0:3, 0:3])
            tvec_guess = W_T_prev[0:3, 3]
            retval, rvec, tvec, inliers = hf.solvePnPRansac(
                objectPoints=known_3d_matched[:, np.newaxis, :].astype(np.float32),
                imagePoints=cur_pts[:, np.newaxis, :].astype(np.float32), cameraMatrix=K, distCoeffs=np.array([]),
                rvec=rvec_guess.copy(), tvec=tvec_guess.copy(), useExtrinsicGuess=True, context=context, solver=pnp_solver)
            tvec = tvec[:, np.newaxis]
            T = hf.transformMatrix(rvec, tvec)
            r, t = T[:3, :3], np.asarray(T[:3, -1]).squeeze()
            prev_pose = Isometry3d(R=np.asarray(r), t=t).inverse().matrix()  # W_T_curr
        elif pnp and len(matches) >= 5:  # main.py:189-204
            known = known_3d[matches.query_idx]
            c_T_w = np.linalg.inv(prev_pose)
            ok, rvec, tvec, _ = hf.solvePnPRansac(known, cur_pts, K, None, hf.Rtorvec(c_T_w[:3, :3]), c_T_w[:3, 3],
                                                  useExtrinsicGuess=True, context=context, seed=k, solver=pnp_solver)
            if ok:
                prev_pose = np.linalg.inv(np.asarray(hf.transformMatrix(rvec, tvec)))
        local_map.AddParentAndPose(parent_id=k - 1, frame_id=k, frame_obj=cur, rel_pose_trans=np.eye(4), pose=prev_pose)
        local_map.AddPointToFrameCorrespondences(point_ids=[point_ids[m[0].queryIdx] for m in matches],
                                                 image_points=cur_pts, descriptors=cur_fts, frame_obj=cur)
        BundleAdjustment(camera, context=context, solver=ba_solver).motionOnlyBundleAdjustement(local_map)
    dt = time.perf_counter() - t0
    return np.stack([local_map.GetFrame(k).GetPose() for k in range(len(frames))]), dt


def gpu_callables(ctx):
    def detect(bgr):
        xy, _, desc = ctx.detect_describe_bgr(bgr, 20, 3000)
        return xy, desc

    def match(q, t):
        mq, mt, _ = ctx.match_ratio(q, t, 0.8)
        return mq, mt

    def ba(*problem):
        return ctx.ba_solve(*problem, huber_delta=HUBER, max_iterations=10)

    return detect, match, ba


def gpu_pnp(ctx):
    def pnp(obj, img, K4, pose0, seed=0):
        return ctx.pnp_ransac(obj, img, K4, pose0, seed=seed)

    return pnp


def bench_frames(ctx, repeats=5):
    """frames/s of the 20-frame ICL-NUIM stream through the host C ABI (PNG decode excluded, H2D copies included), the
    class API and the device-resident period.  Every figure is the MEDIAN of `repeats` runs of the whole period (the
    fastest run is reported beside it as *_max).  Returns (report dict, poses) -- bench.py times the CPU oracle through
    track_sequence() for the comparison."""
    import statistics
    frames, depth0 = load_sequence(20)
    frames = [ctx.pin(f) for f in frames]  # decoded frames live in pinned memory: H2D is a plain DMA
    det, mat, ba = gpu_callables(ctx)
    pnp = gpu_pnp(ctx)
    n = len(frames)

    def timed(fn, reps):
        ts, last = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            last = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), min(ts), last

    track_sequence(det, mat, ba, frames[:4], depth0, pnp=pnp)  # warm-up (allocations, code objects)
    dt, dt_min, (poses, stages, nm) = timed(lambda: track_sequence(det, mat, ba, frames, depth0, pnp=pnp), repeats)
    # the same period through the reference's class API (Frame / Map / FeatureMatcher / BundleAdjustment objects)
    track_sequence_api(frames[:3], depth0, context=ctx)
    api_dt, api_min, (api_poses, _) = timed(lambda: track_sequence_api(frames, depth0, context=ctx), repeats)
    # ... and with main.py:187-204 as it stands (float32 object points, the camera-to-world pose as the solver's guess)
    track_sequence_api(frames[:3], depth0, context=ctx, verbatim=True)
    vb_dt, vb_min, (vb_poses, _) = timed(lambda: track_sequence_api(frames, depth0, context=ctx, verbatim=True), repeats)
    from .map import Map
    Map.use_device_mirror = False  # the same calls with nothing resident: every step uploads its own arrays
    try:
        vb_plain, _ = track_sequence_api(frames, depth0, context=ctx, verbatim=True)
    finally:
        Map.use_device_mirror = True
    # the same period with the map resident on the device (one image upload per frame)
    track_sequence_resident(ctx, frames[:4], depth0)
    res_dt, res_min, (res_poses, _, _) = timed(lambda: track_sequence_resident(ctx, frames, depth0), repeats)
    track_sequence_resident(ctx, frames[:4], depth0, pipelined=True)
    pipe_dt, pipe_min, (pipe_poses, _, _) = timed(lambda: track_sequence_resident(ctx, frames, depth0, pipelined=True), repeats)
    out = {"statistic": "median of %d runs of the 20-frame period (fastest run as *_max)" % repeats,
           "frames_per_s": n / dt, "frames_per_s_max": n / dt_min, "n_frames": n, "seconds": dt,
           "resident_frames_per_s": n / res_dt, "resident_frames_per_s_max": n / res_min,
           "resident_pipelined_frames_per_s": n / pipe_dt, "resident_pipelined_frames_per_s_max": n / pipe_min,
           "resident_pipelined_equals_resident": bool(np.array_equal(pipe_poses, res_poses)),
           "resident_vs_array_path": float(max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(res_poses, poses))),
           "class_api_frames_per_s": n / api_dt, "class_api_frames_per_s_max": n / api_min,
           "class_api_vs_array_path": float(max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(api_poses, poses))),
           "class_api_verbatim_frames_per_s": n / vb_dt, "class_api_verbatim_frames_per_s_max": n / vb_min,
           "class_api_verbatim_vs_nothing_resident": float(max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(vb_poses, vb_plain))),
           "class_api_note": "class_api_*: the reference's call sequence with float64 object points and the previous pose as the "
                             "solver's guess (equals the array path); class_api_verbatim_*: main.py:187-204 as it stands "
                             "(objectPoints.astype(float32), rvec / tvec taken from the camera-to-world pose), compared with the "
                             "same calls when nothing is kept resident",
           "stage_ms_per_frame": {k: v / n * 1e3 for k, v in stages.items()},
           "mean_matches": float(np.mean(nm)), "resolution": "640x480",
           "data": "ICL-NUIM living-room traj3 frames 0-19 (fixtures)",
           "note": "host C-ABI path incl. H2D/D2H copies; detect+describe -> match -> PnP-RANSAC -> motion-only BA per "
                   "frame; map initialised from depth of frame 0"}
    return out, poses
