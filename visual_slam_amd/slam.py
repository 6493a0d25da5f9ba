"""Headless driver: the control flow of the reference's src/v2/main.py:150-348 (tracking loop + key-frame insertion) on
the classes of this package, without viewer, cv2 or g2o.

Every arithmetic stage goes through an injectable back end so that the identical driver runs on the GPU (default) and
on the CPU oracle (tests, bench baseline):
    detect+describe  FeatureExtractor            main.py:181
    match            FeatureMatcher              main.py:185, 244
    motion-only BA   BundleAdjustment            main.py:213-214
    triangulation    helper_functions.triangulate main.py:284
    local BA         BundleAdjustment            main.py:322-323
    PnP-RANSAC       helper_functions.solvePnPRansac  main.py:196-204
    two-view init    helper_functions.estimateEssential / estimateRelativePose  main.py:88-148 (init="two_view")
Initialisation: `init="depth"` (default) back-projects the first frame's keypoints with the dataset's depth image and
applies the normalisation main.py applies after its first BA (everything divided by the median point norm,
LocalBA.py:178-190); `init="two_view"` is main.py:78-148 - essential matrix + recoverPose between frame 0 and the first
later frame with >= 100 matches and >= 90 % cheirality-valid inliers, then local BA with scale=True.  The frames consumed
by the initialisation get the second key frame's pose in the returned trajectory.
main.py:193-194 hands solvePnPRansac the previous camera-to-world pose as if it were world-to-camera; here the guess is
the previous world-to-camera transform (`pnp_guess="w2c"`); `pnp_guess="reference"` reproduces the reference's call,
`pnp_guess=None` skips PnP and starts motion-only BA from the previous pose.
The key-frame rule is main.py:221 with the frame gap as a parameter (20 there).
"""
import copy

import numpy as np

from . import helper_functions as hf
from .LocalBA import BundleAdjustment, Camera, Isometry3d
from .frame import FeatureExtractor, FeatureMatcher, Frame
from .map import Map
from .point import Point

kMaxPeriod = 256  # frames of one device-resident tracking period (see run_sequence.open_period)


class Backends:
    """extractor / matcher objects and factories for BundleAdjustment and triangulate."""

    def __init__(self, context=None, ba_solver=None, extractor=None, matcher=None, triangulate=None, pnp_solver=None,
                 essential_solver=None, recover_solver=None):
        self.extractor = extractor or FeatureExtractor(context=context)
        self.matcher = matcher or FeatureMatcher(context=context)
        self._ctx, self._solver, self._pnp = context, ba_solver, pnp_solver
        self._ess, self._rec = essential_solver, recover_solver
        self.triangulate = triangulate or (lambda P1, P2, x1, x2: hf.triangulate(P1, P2, x1, x2, context=context))

    def pnp(self, obj, img, K, rvec, tvec, seed):
        return hf.solvePnPRansac(obj, img, K, None, rvec, tvec, useExtrinsicGuess=True, context=self._ctx, seed=seed,
                                 solver=self._pnp)

    def essential(self, pts1, pts2, K, essTh, seed=0):
        return hf.estimateEssential(pts1, pts2, K, essTh, context=self._ctx, seed=seed, solver=self._ess)

    def relative_pose(self, E, p1, p2, f1, f2, K):
        return hf.estimateRelativePose(E, p1, p2, f1, f2, K, "Essential", context=self._ctx, solver=self._rec)

    def ba(self, camera):
        return BundleAdjustment(camera, context=self._ctx, solver=self._solver)


def _inv(pose):
    """Inverse of a rigid 4x4: what main.py spells Isometry3d(R=..., t=...).inverse().matrix() (main.py:44-45: R.T, -R.T @ t) --
    the same two products, without the three temporary objects (it runs three times per tracked frame)."""
    pose = np.asarray(pose)
    Rt = pose[0:3, 0:3].T
    m = np.zeros((4, 4))
    m[:3, :3] = Rt
    m[:3, 3] = -Rt @ np.asarray(pose[:3, -1]).squeeze()
    m[3, 3] = 1.0
    return m


def two_view_init(frames, K, camera, be, map, min_matches=100, min_valid=0.9, log=None):
    """main.py:78-148: returns (index of the second key frame, next point id); the map then holds two key frames, the
    triangulated points and has been bundle-adjusted with scale=True."""
    id_frame, id_point = 0, 1
    cur_frame = Frame(frames[0], None, id_frame)
    cur_frame.AddPose(init_pose=np.eye(4))
    cur_frame.SetAsKeyFrame()
    cur_frame.AddParent(None, None)
    kp_prev, features_prev, _ = cur_frame.process_frame(be.extractor)
    map.AddFrame(frame_id=id_frame, frame=cur_frame)
    id_frame += 1
    for i in range(1, len(frames)):
        prev_frame = map.GetFrame(id_frame - 1)
        cur_frame = Frame(frames[i], None, id_frame)
        kp_cur, features_cur, _ = cur_frame.process_frame(be.extractor)
        matches, p1, f1, p2, f2 = be.matcher.match_features(kp_prev, features_prev, kp_cur, features_cur)
        if len(matches) < min_matches:
            continue
        E, inliers, score = be.essential(p1, p2, K, essTh=3.0 / K[0, 0], seed=i)
        sel = inliers[:, 0] == 1
        if sel.sum() < 8:
            continue
        R, t, valid, X, q1, q2, g1, g2 = be.relative_pose(E, p1[sel], p2[sel], f1[sel], f2[sel], K)
        if log:
            log("two-view init, image %d: %d matches, %d inliers, valid fraction %.2f" % (i, len(matches), sel.sum(), valid))
        if valid < min_valid:
            continue
        rel = Isometry3d(R=R, t=np.squeeze(t)).inverse().matrix()
        pose = rel @ prev_frame.GetPose()
        map.AddParentAndPose(parent_id=id_frame - 1, frame_id=id_frame, frame_obj=cur_frame, rel_pose_trans=rel, pose=pose)
        pts = (np.linalg.inv(prev_frame.GetPose()) @ X).T
        pts = pts[:, :3] / np.asarray(pts[:, -1]).reshape(-1, 1)
        for pt, uv1, uv2, ft1, ft2 in zip(pts, q1, q2, g1, g2):
            pt_object = Point(location=pt, id=id_point)
            pt_object.AddFrame(frame=prev_frame, uv=uv1, descriptor=ft1)
            pt_object.AddFrame(frame=cur_frame, uv=uv2, descriptor=ft2)
            map.AddPoint3D(point_id=id_point, point_3d=pt_object)
            id_point += 1
        cur_frame.SetAsKeyFrame()
        be.ba(camera).localBundleAdjustement(map, scale=True)  # main.py:146-148
        return i, id_point
    raise RuntimeError("two-view initialisation failed: no frame pair with enough parallax")


def _new_point_guards(X, w2c1, w2c2, Proj1, Proj2, x1, x2, min_parallax_deg, max_reproj_px):
    """Boolean mask over triangulated points X [n,3]: parallax between the rays from the two camera centres, reprojection error
    in both images.  Not in the reference (run_sequence's guards)."""
    ok = np.ones(len(X), bool)
    if min_parallax_deg is not None:
        c1 = -w2c1[:3, :3].T @ w2c1[:3, 3]
        c2 = -w2c2[:3, :3].T @ w2c2[:3, 3]
        a, b = X - c1, X - c2
        cos = np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
        ok &= np.degrees(np.arccos(np.clip(cos, -1.0, 1.0))) >= min_parallax_deg
    if max_reproj_px is not None:
        Xh = np.c_[X, np.ones(len(X))]
        for P, x in ((Proj1, x1), (Proj2, x2)):
            p = (P @ Xh.T).T
            with np.errstate(divide="ignore", invalid="ignore"):
                err = np.linalg.norm(p[:, :2] / p[:, 2:] - x[:, :2], axis=1)
            ok &= err <= max_reproj_px
    return ok


class _NoLaps:
    """Stands where tools/keyframe_stages.py puts its stage clock: lap(name) after a statement of the key-frame block."""

    @staticmethod
    def lap(name):
        pass


def run_sequence(frames, depth0, K4, backends, keyframe_gap=20, min_tracked=80, max_depth=1.0, log=None,
                 pnp_guess="w2c", init="depth", resident_ctx=None, stages=None, pipelined=True,
                 new_point_min_parallax_deg=None, new_point_max_reproj_px=None, keyframe_ba="reference"):
    """frames: list of BGR images; depth0: metric depth of frames[0]; K4 = (fx, fy, cx, cy).
    resident_ctx: a Context -> the frames between two key frames run on the device-resident tracking period
    (Context.track_begin / track_frame: main.py:181-214 as one call); key-frame insertion stays on the class API.
    pipelined (resident mode): frame i + 1 is submitted to the period BEFORE frame i's result is looked at
    (Context.track_frame_pipelined: its upload, detection and match run beside frame i's PnP and motion-only BA).  Whether
    frame i becomes a key frame is only known from that result, so the submission is a speculation: when frame i does become
    one -- one frame in `keyframe_gap` -- the period ends, the frame in flight is dropped with it and goes to the new period
    afresh.  A tracked frame's key points, descriptors and match lists are needed only if it becomes a key frame
    (main.py:221-236) and are fetched then (Context.track_last_frame), not per frame.  Same results as frame by frame.
    new_point_min_parallax_deg / new_point_max_reproj_px / keyframe_ba: GUARDS THAT THE REFERENCE DOES NOT HAVE (default: off =
    main.py's behaviour).  main.py:291-309 accepts a triangulated point on positive depth (< 1) alone, and main.py:322 frees every
    point in a bundle adjustment whose only gauge is the first pose; on real data either one collapses the map's
    scale within a few key frames.  With the guards a new point also needs the given parallax between its two rays and the given
    reprojection error in both views, and keyframe_ba="poses_only" adjusts the key-frame poses with the points held fixed
    (BundleAdjustment.keyframePoseAdjustement).
    stages: an object with lap(name), called after every statement of the key-frame block (tools/keyframe_stages.py).
    Returns dict(poses [n,4,4] camera-to-world, keyframes [indices], n_points, map, tracked [per frame])."""
    lap = (stages or _NoLaps).lap
    fx, fy, cx, cy = K4
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    camera = Camera(fx, fy, cx, cy)
    be = backends
    map = Map()
    id_frame, id_point = 0, 1
    start = 1
    if init == "two_view":
        start, id_point = two_view_init(frames, K, camera, be, map, log=log)
        id_frame = 2
        start += 1
    else:
        # ---- initialisation from the depth image of frame 0
        cur_frame = Frame(frames[0], None, id_frame)
        cur_frame.AddPose(init_pose=np.eye(4))
        cur_frame.SetAsKeyFrame()
        cur_frame.AddParent(None, None)
        kp0, ft0, _ = cur_frame.process_frame(be.extractor)
        map.AddFrame(frame_id=id_frame, frame=cur_frame)
        z = depth0[kp0[:, 1].astype(int), kp0[:, 0].astype(int)]
        pts = np.stack([(kp0[:, 0] - cx) * z / fx, (kp0[:, 1] - cy) * z / fy, z], 1)
        pts = pts / np.median(np.linalg.norm(pts, axis=1))  # LocalBA.py:178-190 (scale=True): median point norm = 1
        if resident_ctx is not None:  # (the explicit resident period is this project's own driver: the points go in as one batch)
            map.AddPoints3D(range(id_point, id_point + len(pts)), pts, [(cur_frame, kp0, ft0)])
            id_point += len(pts)
        else:                         # main.py's statements
            for X, uv, ft in zip(pts, kp0, ft0):
                pt_object = Point(location=X, id=id_point)
                pt_object.AddFrame(frame=cur_frame, uv=uv, descriptor=ft)
                map.AddPoint3D(point_id=id_point, point_3d=pt_object)
                id_point += 1
        id_frame += 1
    resident = resident_ctx is not None
    last_keyframe = copy.copy(map.GetFrame(frame_id=id_frame - 1))  # main.py:153
    local_map = Map()
    local_map.AddFrame(last_keyframe.GetID(), last_keyframe)
    if not resident:  # the resident period reads the key frame's points straight from the global map (it never edits them)
        local_map.Store3DPoints(map.GetCopyOfPointObjects(last_keyframe.GetID()))
    id_frame_local = id_frame
    loop_idx = start - 1
    all_poses = {0: np.array(map.GetFrame(0).GetPose(), dtype=np.float64)}
    keyframes, tracked, pnp_inliers = [0], [], []
    if init == "two_view":
        for k in range(1, start):
            all_poses[k] = np.array(map.GetFrame(1).GetPose(), dtype=np.float64)
        keyframes.append(start - 1)
    # ---- tracking loop (main.py:173-348)
    period = {"cap": 0, "used": 0}

    def open_period(first_frame):
        """resident mode: the local map of the new period (the last key frame's points) goes to the device once.  The
        period is sized for the frames that can still follow, at most kMaxPeriod: the device keeps cap x points
        observation slots and reads back cap + 1 camera records per frame, so it must not be sized by the sequence.  A
        period that fills up ends with a forced key frame (the reference's local map grows without bound, main.py:210)."""
        _, ft, xyz, ids = map.GetImagePointsWithFrameID(last_keyframe.GetID())
        period["cap"] = max(1, min(len(frames) - first_frame, kMaxPeriod))
        period["used"] = 0
        resident_ctx.track_begin(xyz, ft, last_keyframe.GetPose(), K4, max_frames=period["cap"],
                                 pnp_iterations=100 if pnp_guess is not None else 0)
        return ids, len(xyz)

    in_flight = -1  # resident + pipelined: index of the frame submitted to the period whose result has not been taken yet
    if resident:
        point_IDs, n_known = open_period(start)
    for i in range(start, len(frames)):
        cur_frame = Frame(frames[i], None, id_frame_local)
        if resident:
            # main.py:181-214 as one call on the device-resident period (vs_track_frame / vs_track_frame_pipelined); the frame's
            # key points and match lists stay on the device unless the frame becomes a key frame (below)
            if pipelined:
                if in_flight != i:  # first frame of a period
                    resident_ctx.track_frame_pipelined(frames[i], seed=i, want_matches=False)
                # ... unless this frame is due to become a key frame anyway (main.py:221's first condition holds before the frame is
                # tracked): then the frame in flight would be dropped with the period, and the device would work on it while the
                # key frame's bundle adjustment waits.  A guess that only decides whether work is wasted, never a result.
                due = i - loop_idx > keyframe_gap
                nxt = i + 1 if i + 1 < len(frames) and period["used"] + 2 <= period["cap"] and not due else -1
                r = resident_ctx.track_frame_pipelined(frames[nxt] if nxt >= 0 else None, seed=nxt, want_matches=False)
                in_flight = nxt
            else:
                r = resident_ctx.track_frame(frames[i], seed=i, want_matches=False)
            period["used"] += 1
            n_cur = r["n_matches"]
            W_T_cur = r["poses"][-1]
            pnp_inliers.append(r["pnp_inliers"])
        else:
            kp_cur, features_cur, _ = cur_frame.process_frame(be.extractor)
            kp_prev, features_prev, known_3d, point_IDs = local_map.GetImagePointsWithFrameID(last_keyframe.GetID())
            n_known = len(known_3d)
            matches, _, _, curMatchedPoints, curMatchedFeatures = be.matcher.match_features(kp_prev, features_prev, kp_cur,
                                                                                             features_cur)
            if hasattr(matches, "query_idx"):  # MatchList: the same gathers as main.py:187-188's loops, as two fancy indexings
                known_3d_matched_ids = np.asarray(point_IDs)[matches.query_idx]
                known_3d_matched = np.asarray(known_3d).reshape(-1, 3)[matches.query_idx]
            else:
                known_3d_matched_ids = [point_IDs[m[0].queryIdx] for m in matches]
                known_3d_matched = np.array([known_3d[m[0].queryIdx] for m in matches]).reshape(-1, 3)
            # pose from PnP-RANSAC with the previous frame as extrinsic guess (main.py:191-204)
            W_T_prev = np.array(local_map.GetFrame(id_frame_local - 1).GetPose(), dtype=np.float64)
            W_T_curr = W_T_prev.copy()
            prev_T_W = _inv(W_T_prev)
            if pnp_guess is not None and len(known_3d_matched) >= 5:
                guess = W_T_prev if pnp_guess == "reference" else prev_T_W
                retval, rvec, tvec, inl = be.pnp(known_3d_matched, curMatchedPoints, K, hf.Rtorvec(guess[:3, :3]),
                                                 np.array(guess[:3, 3]), seed=i)
                if retval:
                    T = hf.transformMatrix(rvec, tvec)
                    W_T_curr = np.asarray(_inv(np.asarray(T)))
                pnp_inliers.append(len(inl))
            else:
                pnp_inliers.append(0)
            RelativePoseTransformation = prev_T_W @ W_T_curr
            local_map.AddParentAndPose(parent_id=id_frame_local - 1, frame_id=id_frame_local, frame_obj=cur_frame,
                                       rel_pose_trans=RelativePoseTransformation, pose=W_T_curr)
            local_map.AddPointToFrameCorrespondences(point_ids=known_3d_matched_ids, image_points=curMatchedPoints,
                                                     descriptors=curMatchedFeatures, frame_obj=cur_frame)
            be.ba(camera).motionOnlyBundleAdjustement(local_map, scale=False, save=True)
            W_T_cur = np.array(local_map.GetFrame(id_frame_local).GetPose())
            n_cur = len(curMatchedPoints)
        all_poses[i] = np.array(W_T_cur)
        tracked.append(n_cur)
        # key-frame rule (main.py:221)
        period_full = resident and period["used"] >= period["cap"] and i + 1 < len(frames)
        if period_full or ((i - loop_idx > keyframe_gap or n_cur < min_tracked) and (n_cur < 0.9 * n_known)):
            lap("(tracking since the last key frame)")
            loop_idx = i
            cur_frame.SetAsKeyFrame()
            W_T_prev_key = map.GetFrame(id_frame - 1).GetPose()
            W_T_cur_key = W_T_cur
            if resident:
                lf = resident_ctx.track_last_frame()  # this frame's key points, descriptors and matches: needed now
                resident_ctx.track_end()                # (a frame in flight is dropped with the period: it goes to the next one)
                in_flight = -1
                cur_frame.keypoints, cur_frame.features = lf["xy"], lf["desc"]
                known_3d_matched_ids = np.asarray(point_IDs)[lf["match_q"]]
                curMatchedPoints, curMatchedFeatures = lf["xy"][lf["match_t"]], lf["desc"][lf["match_t"]]
                cur_frame.AddPose(W_T_cur_key)
                cur_frame.AddID(id_frame)
            cur_frame.ClearParent()
            map.AddParentAndPose(parent_id=id_frame - 1, frame_id=id_frame, frame_obj=cur_frame,
                                 rel_pose_trans=_inv(W_T_prev_key) @ W_T_cur_key, pose=W_T_cur_key)
            lap("track_end + AddParentAndPose")
            map.AddPointToFrameCorrespondences(point_ids=known_3d_matched_ids, image_points=curMatchedPoints,
                                               descriptors=curMatchedFeatures, frame_obj=cur_frame)
            lap("AddPointToFrameCorrespondences")
            if id_frame >= 6 and id_frame % 4 == 0:
                map.DiscardOutlierMapPoints(n_visible_frames=3)
            lap("DiscardOutlierMapPoints")
            # unmatched keypoints of the previous key frame against the new one (main.py:237-244)
            image_points_already_in_map = map.GetImagePointsWithFrameID(id_frame - 1)[0]
            lap("GetImagePointsWithFrameID(previous key frame)")
            kp1 = map.GetFrame(id_frame - 1).GetKeyPoints()
            desc1 = map.GetFrame(id_frame - 1).GetFeatures()
            idx = hf.GetListDiff(kp1, image_points_already_in_map)
            lap("GetListDiff")
            kp1, desc1 = kp1[idx], desc1[idx]
            lap("kp1[idx], desc1[idx]")
            n_new = 0
            if len(kp1) >= 1 and len(map.GetFrame(id_frame).GetKeyPoints()) >= 2:
                _, last_kf_pts, last_kf_fts, cur_kf_pts, cur_kf_fts = be.matcher.match_features(
                    kp1=kp1, desc1=desc1, kp2=map.GetFrame(id_frame).GetKeyPoints(),
                    desc2=map.GetFrame(id_frame).GetFeatures())
                lap("match_features(previous key frame's unmatched, new key frame)")
                if len(last_kf_pts):
                    p1 = _inv(map.GetFrame(id_frame - 1).GetPose())
                    p2 = _inv(map.GetFrame(id_frame).GetPose())
                    Proj1 = hf.CameraProjectionMatrix2(Pose=p1, K=K)
                    Proj2 = hf.CameraProjectionMatrix2(Pose=p2, K=K)
                    x1 = hf.MakeHomogeneous(last_kf_pts)
                    x2 = hf.MakeHomogeneous(cur_kf_pts)
                    lap("projection matrices, MakeHomogeneous")
                    new_pts = np.array(be.triangulate(Proj1, Proj2, x1, x2), dtype=np.float64)
                    lap("triangulate")
                    new_pts /= new_pts[:, 3:]
                    proj1 = p1 @ new_pts.T
                    proj2 = p2 @ new_pts.T
                    new_pts = new_pts[:, :3]
                    ok = (proj1[2] > 0) & (proj2[2] > 0) & (proj2[2] < max_depth) & (proj1[2] < max_depth)
                    if new_point_min_parallax_deg is not None or new_point_max_reproj_px is not None:
                        ok &= _new_point_guards(new_pts, p1, p2, Proj1, Proj2, x1, x2, new_point_min_parallax_deg, new_point_max_reproj_px)
                    good = np.where(ok)[0]
                    lap("cheirality filter")
                    if resident:  # main.py:312-318 as one batch (Map.AddPoints3D leaves the map as the loop below does)
                        n_new = len(good)
                        map.AddPoints3D(range(id_point, id_point + n_new), new_pts[good],
                                        [(map.GetFrame(id_frame - 1), last_kf_pts[good], last_kf_fts[good]),
                                         (map.GetFrame(id_frame), cur_kf_pts[good], cur_kf_fts[good])])
                        id_point += n_new
                    else:
                        for pt, uv1, uv2, ft1, ft2 in zip(new_pts[good], last_kf_pts[good], cur_kf_pts[good],
                                                          last_kf_fts[good], cur_kf_fts[good]):
                            pt_object = Point(location=pt, id=id_point)
                            pt_object.AddFrame(frame=map.GetFrame(id_frame - 1), uv=uv1, descriptor=ft1)
                            pt_object.AddFrame(frame=map.GetFrame(id_frame), uv=uv2, descriptor=ft2)
                            map.AddPoint3D(point_id=id_point, point_3d=pt_object)
                            id_point += 1
                            n_new += 1
                    lap("Point / AddFrame x2 / AddPoint3D per new point")
            if keyframe_ba == "poses_only":
                be.ba(camera).keyframePoseAdjustement(map)
            else:
                be.ba(camera).localBundleAdjustement(map)  # main.py:322-323
            lap("localBundleAdjustement")
            all_poses[i] = np.array(map.GetFrame(id_frame).GetPose())
            keyframes.append(i)
            if log:
                log("key frame at image %d: %d new points, map has %d points" % (i, n_new, len(map.points_3d)))
            last_keyframe = copy.copy(map.GetFrame(frame_id=id_frame))
            local_map = Map()
            local_map.AddFrame(last_keyframe.GetID(), last_keyframe)
            id_frame += 1
            id_frame_local = id_frame
            lap("copy(key frame), new local Map")
            if resident:
                point_IDs, n_known = open_period(i + 1)
                lap("open_period (GetImagePointsWithFrameID + track_begin)")
            else:
                local_map.Store3DPoints(map.GetCopyOfPointObjects(last_keyframe.GetID()))
                lap("GetCopyOfPointObjects + Store3DPoints")
        else:
            id_frame_local += 1
    if resident:
        resident_ctx.track_end()
    poses = np.stack([all_poses[i] for i in range(len(frames))])
    return dict(poses=poses, keyframes=keyframes, n_points=len(map.points_3d), map=map, tracked=tracked,
                pnp_inliers=pnp_inliers)
