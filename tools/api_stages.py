"""dev tool: the class-API tracking period (harness.track_sequence_api's loop, statement by statement) with a wall-clock
accumulator around every statement -- where the host time of the unmodified main.py:181-214 call sequence goes, without a
profiler's per-call overhead."""
import _env  # noqa: F401
import time
from collections import OrderedDict

import numpy as np

from visual_slam_amd import Context, harness
from visual_slam_amd import helper_functions as hf
from visual_slam_amd.LocalBA import BundleAdjustment, Camera
from visual_slam_amd.frame import FeatureExtractor, FeatureMatcher, Frame
from visual_slam_amd.map import Map
from visual_slam_amd.point import Point
from visual_slam_amd.workloads import ICL_NUIM_K

ctx = Context(0)
frames, depth0 = harness.load_sequence(20)
frames = [ctx.pin(f) for f in frames]
acc = OrderedDict()


def lap(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


def run():
    extractor, matcher = FeatureExtractor(context=ctx), FeatureMatcher(context=ctx)
    camera = Camera(*ICL_NUIM_K)
    fx, fy, cx, cy = ICL_NUIM_K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    t = time.perf_counter()
    key = Frame(frames[0], None, 0)
    key.AddPose(np.eye(4))
    key.SetAsKeyFrame()
    kp0, ft0, _ = key.process_frame(extractor)
    local_map = Map()
    local_map.AddFrame(0, key)
    t = lap("period: key frame", t)
    for i, (X, uv, d) in enumerate(zip(harness.backproject(kp0, depth0), kp0, ft0)):
        pt = Point(location=X, id=i + 1)
        pt.AddFrame(frame=key, uv=uv, descriptor=d)
        local_map.AddPoint3D(point_id=i + 1, point_3d=pt)
    t = lap("period: 595 x Point / AddFrame / AddPoint3D", t)
    for k in range(1, len(frames)):
        cur = Frame(frames[k], None, k)
        t = lap("Frame()", t)
        kp_cur, ft_cur, _ = cur.process_frame(extractor)
        t = lap("process_frame (front half on the GPU)", t)
        kp_prev, ft_prev, known_3d, point_ids = local_map.GetImagePointsWithFrameID(0)
        t = lap("GetImagePointsWithFrameID", t)
        matches, _, _, cur_pts, cur_fts = matcher.match_features(kp_prev, ft_prev, kp_cur, ft_cur)
        t = lap("match_features", t)
        prev_pose = np.asarray(local_map.GetFrame(k - 1).GetPose(), np.float64)
        known = known_3d[matches.query_idx]
        c_T_w = np.linalg.inv(prev_pose)
        rv = hf.Rtorvec(c_T_w[:3, :3])
        t = lap("pose inverse, Rtorvec, gather", t)
        ok, rvec, tvec, _ = hf.solvePnPRansac(known, cur_pts, K, None, rv, c_T_w[:3, 3], useExtrinsicGuess=True, context=ctx, seed=k)
        t = lap("solvePnPRansac (PnP on the GPU)", t)
        if ok:
            prev_pose = np.linalg.inv(np.asarray(hf.transformMatrix(rvec, tvec)))
        t = lap("transformMatrix + inverse", t)
        local_map.AddParentAndPose(parent_id=k - 1, frame_id=k, frame_obj=cur, rel_pose_trans=np.eye(4), pose=prev_pose)
        t = lap("AddParentAndPose", t)
        ids = [point_ids[m[0].queryIdx] for m in matches]
        t = lap("[point_ids[m[0].queryIdx] for m in matches]", t)
        local_map.AddPointToFrameCorrespondences(point_ids=ids, image_points=cur_pts, descriptors=cur_fts, frame_obj=cur)
        t = lap("AddPointToFrameCorrespondences", t)
        BundleAdjustment(camera, context=ctx).motionOnlyBundleAdjustement(local_map)
        t = lap("motionOnlyBundleAdjustement (rest of the BA)", t)


def timed(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0)
    setattr(obj, name, g)


timed(ctx, "track_back_end", "  (inside: ctx.track_back_end)")
timed(ctx, "track_back_begin", "  (inside: ctx.track_back_begin)")
timed(ctx, "track_front", "  (inside: ctx.track_front)")
from visual_slam_amd import map as _map  # noqa: E402
timed(_map._PeriodMirror, "solve", "  (inside: _PeriodMirror.solve incl. track_back_end)")
timed(_map._PeriodMirror, "speculate_back", "  (inside: _PeriodMirror.speculate_back incl. track_back_begin)")
timed(_map._PeriodMirror, "speculate_front", "  (inside: _PeriodMirror.speculate_front incl. track_front)")
for _ in range(3):
    run()
acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    run()
total = (time.perf_counter() - t0) / N / 20 * 1e6
print("class API, instrumented: %.1f us per frame" % total)
for k, v in acc.items():
    print("  %-52s %7.1f us per frame" % (k, v / N / 20 * 1e6))
ctx.close()
