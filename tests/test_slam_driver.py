"""The headless driver (visual_slam_amd/slam.py = the control flow of the reference's main.py:150-348) end to end on
the 20 ICL-NUIM fixture frames: BASELINE.json configs[0] (traj3 instead of traj0).  CPU: the driver on
the oracle back ends.  GPU: the same driver on the HIP back ends must take the same decisions (key frames, new
points) and end within the BA contract of the CPU run."""
import os

import numpy as np
import pytest

from visual_slam_amd import harness, slam
from visual_slam_amd.frame import MatchList
from visual_slam_amd.workloads import ICL_NUIM_K


class OracleExtractor:
    def __init__(self, oracle):
        self.o = oracle

    def compute_features(self, img):
        xy, _, desc = self.o.detect_describe_bgr(img, 20, 3000)
        return xy, desc


class OracleMatcher:
    def __init__(self, oracle):
        self.o = oracle

    def match_features(self, kp1, desc1, kp2, desc2, ratio=0.8):
        mq, mt, md = self.o.match_ratio(desc1, desc2, ratio)
        return MatchList(mq, mt, md), np.asarray(kp1)[mq], np.asarray(desc1)[mq], np.asarray(kp2)[mt], np.asarray(desc2)[mt]


def oracle_backends(oracle):
    from oracle import np_reference
    return slam.Backends(ba_solver=oracle.ba_solve, extractor=OracleExtractor(oracle), matcher=OracleMatcher(oracle),
                         triangulate=np_reference.triangulate, pnp_solver=oracle.pnp_ransac,
                         essential_solver=oracle.essential_ransac, recover_solver=oracle.recover_pose)


def _run(be, n=20, gap=4):
    frames, depth0 = harness.load_sequence(n)
    return slam.run_sequence(frames, depth0, ICL_NUIM_K, be, keyframe_gap=gap, min_tracked=80)


def test_driver_on_the_oracle_back_ends(oracle):
    r = _run(oracle_backends(oracle))
    assert r["keyframes"][0] == 0 and len(r["keyframes"]) >= 3          # key frames were inserted ...
    assert r["n_points"] > 595                                          # ... and triangulation added map points
    assert min(r["tracked"]) > 100
    assert min(r["pnp_inliers"]) > 0.8 * min(r["tracked"])                # PnP-RANSAC found a model every frame
    step = np.linalg.norm(np.diff(r["poses"][:, :3, 3], axis=0), axis=1)
    assert step.max() < 0.05                                            # millimetre motion, no jumps
    for P in r["poses"]:
        assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-9)
    m = r["map"]
    assert all(p.GetNVisibleFrames() >= 1 for p in m.points_3d.values()) and list(m.frames)[0] == 0


def test_new_point_guards():
    """slam._new_point_guards (NOT in the reference): parallax between the two rays and reprojection error in both views."""
    K = np.array([[ICL_NUIM_K[0], 0, ICL_NUIM_K[2]], [0, ICL_NUIM_K[1], ICL_NUIM_K[3]], [0, 0, 1.0]])
    w2c1, w2c2 = np.eye(4), np.eye(4)
    w2c2[0, 3] = -0.1                                   # second camera 10 cm to the right
    P1, P2 = K @ w2c1[:3], K @ w2c2[:3]
    X = np.array([[0.0, 0.0, 1.0], [0.2, -0.1, 4.0], [0.0, 0.0, 20.0]])   # parallax 5.7, 1.4 and 0.29 degrees
    Xh = np.c_[X, np.ones(3)]
    x1 = (P1 @ Xh.T).T
    x2 = (P2 @ Xh.T).T
    x1, x2 = x1 / x1[:, 2:], x2 / x2[:, 2:]
    g = slam._new_point_guards
    assert g(X, w2c1, w2c2, P1, P2, x1, x2, 1.0, None).tolist() == [True, True, False]
    assert g(X, w2c1, w2c2, P1, P2, x1, x2, 2.0, None).tolist() == [True, False, False]
    assert g(X, w2c1, w2c2, P1, P2, x1, x2, None, 2.0).tolist() == [True, True, True]
    x2b = x2.copy()
    x2b[1, 0] += 3.0                                    # three pixels off in the second view
    assert g(X, w2c1, w2c2, P1, P2, x1, x2b, None, 2.0).tolist() == [True, False, True]
    assert g(X, w2c1, w2c2, P1, P2, x1, x2b, 1.0, 2.0).tolist() == [True, False, False]


def test_guarded_driver_on_the_oracle_back_ends(oracle):
    """The driver's guards (new points need parallax and a small reprojection error; key-frame poses adjusted with the points held
    fixed) on the 20 fixture frames: consecutive key frames are millimetres apart, so no new point passes the parallax guard --
    the map keeps the initialisation's points, untouched by the key-frame adjustment, and the trajectory stays smooth."""
    frames, depth0 = harness.load_sequence(20)
    kw = dict(keyframe_gap=4, min_tracked=80, new_point_min_parallax_deg=1.0, new_point_max_reproj_px=2.0, keyframe_ba="poses_only")
    r = slam.run_sequence(frames, depth0, ICL_NUIM_K, oracle_backends(oracle), **kw)
    ref = slam.run_sequence(frames, depth0, ICL_NUIM_K, oracle_backends(oracle), keyframe_gap=4, min_tracked=80)
    assert r["keyframes"] == ref["keyframes"] and r["n_points"] == 595 < ref["n_points"]
    X = np.array([p.location_3d for p in r["map"].points_3d.values()])
    X0 = np.array([p.location_3d for p in list(ref["map"].points_3d.values())[:595]])
    assert not np.allclose(X, X0)                                           # (the reference's BA moved its points; this one did not)
    step = np.linalg.norm(np.diff(r["poses"][:, :3, 3], axis=0), axis=1)
    assert step.max() < 0.05


@pytest.mark.gpu
def test_guarded_driver_gpu_equals_oracle(vs, oracle):
    frames, depth0 = harness.load_sequence(20)
    kw = dict(keyframe_gap=4, min_tracked=80, new_point_min_parallax_deg=0.02, new_point_max_reproj_px=2.0, keyframe_ba="poses_only")
    g = slam.run_sequence(frames, depth0, ICL_NUIM_K, slam.Backends(context=vs), resident_ctx=vs, **kw)
    c = slam.run_sequence(frames, depth0, ICL_NUIM_K, oracle_backends(oracle), **kw)
    assert g["keyframes"] == c["keyframes"] and g["tracked"] == c["tracked"] and g["n_points"] == c["n_points"] > 595
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], c["poses"])) < 1e-4


@pytest.mark.gpu
def test_driver_gpu_equals_oracle(vs, oracle):
    g = _run(slam.Backends(context=vs))
    c = _run(oracle_backends(oracle))
    assert g["keyframes"] == c["keyframes"] and g["tracked"] == c["tracked"] and g["n_points"] == c["n_points"]
    assert g["pnp_inliers"] == c["pnp_inliers"]
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], c["poses"]))
    assert rel < 1e-4, rel
    pg = np.array([p.location_3d for p in g["map"].points_3d.values()])
    pc = np.array([p.location_3d for p in c["map"].points_3d.values()])
    # new points come out of local BA started from PnP poses that agree to ~1e-9: low-parallax points amplify that
    assert np.abs(pg - pc).max() < 1e-5, np.abs(pg - pc).max()


@pytest.mark.gpu
def test_driver_with_the_resident_tracking_period_equals_the_class_api_driver(vs):
    """run_sequence(resident_ctx=...): the frames between key frames go through vs_track_frame, key frames through the
    class API - same decisions, same map, same trajectory."""
    frames, depth0 = harness.load_sequence(20)
    a = slam.run_sequence(frames, depth0, ICL_NUIM_K, slam.Backends(context=vs), keyframe_gap=4, min_tracked=80)
    b = slam.run_sequence(frames, depth0, ICL_NUIM_K, slam.Backends(context=vs), keyframe_gap=4, min_tracked=80,
                          resident_ctx=vs)
    assert a["keyframes"] == b["keyframes"] and a["tracked"] == b["tracked"] and a["n_points"] == b["n_points"]
    assert a["pnp_inliers"] == b["pnp_inliers"]
    assert max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(a["poses"], b["poses"])) < 1e-7
    pa = np.array([p.location_3d for p in a["map"].points_3d.values()])
    pb = np.array([p.location_3d for p in b["map"].points_3d.values()])
    assert np.abs(pa - pb).max() < 1e-5    # low-parallax points amplify pose differences of ~1e-9 (see above)


@pytest.mark.gpu
def test_pipelined_resident_driver_equals_frame_by_frame(vs):
    """run_sequence(resident_ctx=..., pipelined=True) submits frame i + 1 before it knows whether frame i becomes a key frame
    (the submission is dropped and repeated when it does) and fetches a frame's key points and matches only for key frames:
    bit for bit the trajectory, decisions and map of the frame-by-frame form."""
    frames, depth0 = harness.load_sequence(20)
    # (gap, min_tracked): key frames that are due (no frame in flight when they come) and, with min_tracked out of reach, key
    # frames decided by the tracked count alone -- every one of those drops the frame submitted ahead of it
    for gap, min_tracked in ((4, 80), (2, 80), (20, 80), (100, 10 ** 6)):
        kw = dict(keyframe_gap=gap, min_tracked=min_tracked, resident_ctx=vs)
        a = slam.run_sequence(frames, depth0, ICL_NUIM_K, slam.Backends(context=vs), pipelined=False, **kw)
        b = slam.run_sequence(frames, depth0, ICL_NUIM_K, slam.Backends(context=vs), pipelined=True, **kw)
        if min_tracked > 1000:
            assert len(a["keyframes"]) >= 3
        assert a["keyframes"] == b["keyframes"] and a["tracked"] == b["tracked"] and a["pnp_inliers"] == b["pnp_inliers"]
        assert np.array_equal(a["poses"], b["poses"]) and a["n_points"] == b["n_points"]
        pa = np.array([p.location_3d for p in a["map"].points_3d.values()])
        pb = np.array([p.location_3d for p in b["map"].points_3d.values()])
        assert np.array_equal(pa, pb)
        for fa, fb in zip(a["map"].frames.values(), b["map"].frames.values()):
            assert np.array_equal(fa.GetKeyPoints(), fb.GetKeyPoints()) and np.array_equal(fa.GetFeatures(), fb.GetFeatures())


@pytest.mark.gpu
def test_track_last_frame_hands_out_what_the_frame_call_would_have(vs):
    frames, depth0 = harness.load_sequence(5)
    xy0, _, d0 = vs.detect_describe_bgr(frames[0], 20, 3000)
    X = harness.backproject(xy0, depth0)
    from visual_slam_amd.context import VsError
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=8)
    with pytest.raises(VsError):
        vs.track_last_frame()                                      # nothing handed out yet
    full = vs.track_frame(frames[1], seed=1, want_keypoints=True, want_matches=True)
    late = vs.track_last_frame()
    for k in ("xy", "desc", "match_q", "match_t"):
        assert np.array_equal(full[k], late[k]), k
    assert late["n_matches"] == full["n_matches"] == len(late["match_q"]) and late["n_keypoints"] == len(late["xy"])
    # pipelined: frame 2's arrays while frame 3 is in flight
    assert vs.track_frame_pipelined(frames[2], seed=2, want_matches=False) is None
    r2 = vs.track_frame_pipelined(frames[3], seed=3, want_matches=False)
    l2 = vs.track_last_frame()
    xy2, _, desc2 = vs.detect_describe_bgr(frames[2], 20, 3000)
    mq, mt, _ = vs.match_ratio(d0, desc2, 0.8)
    assert np.array_equal(l2["xy"], xy2) and np.array_equal(l2["desc"], desc2) and r2["n_matches"] == len(mq)
    assert np.array_equal(l2["match_q"], mq) and np.array_equal(l2["match_t"], mt)
    vs.track_end()                                                 # (drops frame 3)
    with pytest.raises(VsError):
        vs.track_last_frame()


def _two_view(be):
    """main.py:78-148 on the two committed frames with enough parallax (ICL-NUIM traj3 images 0 and 150, ~0.45 m and 33
    degrees apart; consecutive frames are millimetres apart and cannot initialise, as in the reference)."""
    import os
    from visual_slam_amd.LocalBA import Camera
    from visual_slam_amd.frame import imread
    from visual_slam_amd.map import Map
    g = os.path.join(os.path.dirname(__file__), "golden", "icl_nuim", "rgb")
    frames = [imread(os.path.join(g, "0.png")), imread(os.path.join(g, "150.png"))]
    K = np.array([[ICL_NUIM_K[0], 0, ICL_NUIM_K[2]], [0, ICL_NUIM_K[1], ICL_NUIM_K[3]], [0, 0, 1.0]])
    m = Map()
    i, next_id = slam.two_view_init(frames, K, Camera(*ICL_NUIM_K), be, m, min_valid=0.75)
    return m, i, next_id


def test_two_view_initialisation_on_the_oracle_back_ends(oracle):
    from visual_slam_amd import dataset
    m, i, next_id = _two_view(oracle_backends(oracle))
    assert i == 1 and len(m.frames) == 2 and len(m.points_3d) == next_id - 1 >= 50
    assert all(p.GetNVisibleFrames() == 2 for p in m.points_3d.values())
    P1 = np.asarray(m.GetFrame(1).GetPose())
    assert np.allclose(P1[:3, :3] @ P1[:3, :3].T, np.eye(3), atol=1e-9)
    X = np.array([p.location_3d for p in m.points_3d.values()])
    assert abs(np.median(np.linalg.norm(X, axis=1)) - 1.0) < 1e-9          # scale=True: median point norm 1
    # against the data set's ground truth (lines 1 and 151): direction of motion and amount of rotation
    g = os.path.join(os.path.dirname(__file__), "golden", "icl_nuim")
    _, g0 = dataset.read_trajectory(os.path.join(g, "traj3.gt.freiburg.head20"))
    _, g1 = dataset.read_trajectory(os.path.join(g, "traj3.gt.freiburg.line151"))
    rel = np.linalg.inv(g0[0]) @ g1[0]
    from scipy.spatial.transform import Rotation
    ang_gt = np.linalg.norm(Rotation.from_matrix(rel[:3, :3]).as_rotvec())
    ang = np.linalg.norm(Rotation.from_matrix(P1[:3, :3]).as_rotvec())
    assert abs(ang - ang_gt) < np.radians(5), (np.degrees(ang), np.degrees(ang_gt))


@pytest.mark.gpu
def test_two_view_initialisation_gpu_equals_oracle(vs, oracle):
    g, _, gn = _two_view(slam.Backends(context=vs))
    c, _, cn = _two_view(oracle_backends(oracle))
    assert gn == cn
    assert np.linalg.norm(g.GetFrame(1).GetPose() - c.GetFrame(1).GetPose()) < 1e-6
