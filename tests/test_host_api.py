"""CPU: the Python mirror of src/v2 (Frame / Point / Map / BundleAdjustment / FeatureMatcher plumbing) -- host logic
only; the numeric back end is injected (the oracle) because no GPU exists here."""
import numpy as np
import pytest

from visual_slam_amd.LocalBA import BundleAdjustment, Camera, Isometry3d
from visual_slam_amd.frame import DMatch, Frame, MatchList
from visual_slam_amd.map import Map
from visual_slam_amd.point import Point
from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload
from oracle.ref_graph import RefLoopBundleAdjustment


def _frame(i, pose=None, key=False):
    f = Frame(np.zeros((4, 4, 3), np.uint8), None, i)
    if pose is not None:
        f.AddPose(pose)
    if key:
        f.SetAsKeyFrame()
    return f


def test_frame_point_map_semantics():
    m = Map()
    f0, f1 = _frame(0, np.eye(4), key=True), _frame(1, np.eye(4))
    m.AddFrame(0, f0)
    with pytest.raises(Exception, match="Duplicate frame"):
        m.AddFrame(0, f0)
    m.AddParentAndPose(parent_id=0, frame_id=1, frame_obj=f1, rel_pose_trans=np.eye(4), pose=np.eye(4))
    assert list(f1.GetParentIDs()) == [0] and m.GetFrame(1) is f1 and not f1.IsKeyFrame() and f0.IsKeyFrame()
    for pid in (1, 2, 3):
        p = Point(np.array([pid, 0.0, 5.0]), pid)
        p.AddFrame(f0, np.array([10.0 * pid, 20.0]), np.full(32, pid, np.uint8))
        m.AddPoint3D(pid, p)
    with pytest.raises(Exception, match="Duplicate point3d"):
        m.AddPoint3D(1, Point(np.zeros(3), 1))
    m.AddPointToFrameCorrespondences([1, 3], np.array([[1.0, 2.0], [3.0, 4.0]]), np.zeros((2, 32), np.uint8), f1)
    uv, desc, xyz, ids = m.GetImagePointsWithFrameID(1)
    assert ids.tolist() == [1, 3] and uv.tolist() == [[1.0, 2.0], [3.0, 4.0]] and xyz.shape == (2, 3)
    assert m.GetPointsVisibleToFrames([0, 1]) == [1, 3]
    assert m.GetPoint(2).GetNVisibleFrames() == 1 and m.GetPoint(1).IsVisibleTo(1) and not m.GetPoint(2).IsVisibleTo(1)
    assert m.GetPoint(1).GetImagePoint(7) is None
    cp = m.GetCopyOfPointObjects(1)
    assert sorted(cp) == [1, 3] and list(cp[1].frames) == [1] and cp[1] is not m.GetPoint(1)
    cp[1].UpdatePoint(np.zeros(3))
    assert m.GetPoint(1).Get3dPoint()[0] == 1.0  # the copy does not alias the map's point
    m.DiscardOutlierMapPoints(n_visible_frames=2)
    assert sorted(m.points_3d) == [1, 3]
    with pytest.raises(Exception, match="No frame yet added"):
        m.UpdatePose(np.eye(4), 9)
    with pytest.raises(Exception, match="No point yet added"):
        m.UpdatePoint3D(np.zeros(3), 2)
    assert m.GetAll3DPoints().shape == (2, 3) and len(m.GetAllPoses()) == 2
    assert m.Get3DPointsWithIDs([3]).tolist() == [[3.0, 0.0, 5.0]]


def test_matchlist_behaves_like_the_reference_list():
    ml = MatchList([4, 9], [7, 1], [12, 30])
    assert len(ml) == 2 and ml[1][0].trainIdx == 1 and ml[-1][0].queryIdx == 9 and isinstance(ml[0][0], DMatch)
    assert [m[0].queryIdx for m in ml] == [4, 9] and ml[0][0].distance == 12.0
    with pytest.raises(IndexError):
        ml[2]


def test_match_rows_built_in_c_equal_the_python_stand_in():
    """frame.py takes DMatch / rows() from visual_slam_amd._rows (CPython C API, compiled by build()) and falls back to its
    Python classes; both must hand out the same list of one-element lists with the same field values and repr."""
    from visual_slam_amd import frame as fr
    _rows = pytest.importorskip("visual_slam_amd._rows")
    r = np.random.default_rng(3)
    q, t, d = (r.integers(0, 5000, 400).astype(np.int32) for _ in range(3))
    a, b = _rows.rows(q, t, d), fr._py_rows(q, t, d)
    assert fr.DMatch is _rows.DMatch and len(a) == len(b) == 400 and all(len(x) == 1 for x in a)
    for x, y in zip(a, b):
        assert (x[0].queryIdx, x[0].trainIdx, x[0].imgIdx, x[0].distance) == (y[0].queryIdx, y[0].trainIdx, y[0].imgIdx, y[0].distance)
        assert repr(x[0]) == repr(y[0]) and isinstance(x[0].distance, float) and isinstance(x[0].queryIdx, int)
    assert _rows.rows(q[:0], t[:0], d[:0]) == []
    with pytest.raises(ValueError):
        _rows.rows(q, t[:-1], d)
    m = _rows.DMatch(queryIdx=3, trainIdx=4, distance=2.5)
    assert (m.queryIdx, m.trainIdx, m.imgIdx, m.distance) == (3, 4, 0, 2.5)
    ml = MatchList(q, t, d)
    assert [m[0].trainIdx for m in ml] == t.tolist() and next(iter(ml)) is next(iter(ml))  # a real list underneath


def test_rotation_vector_helpers_round_trip():
    """Rtorvec / transformMatrix (helper_functions.py:269-278) in plain float arithmetic: against scipy over small, ordinary
    and near-pi rotations; the 4x4 keeps the reference's np.matrix type."""
    from scipy.spatial.transform import Rotation
    from visual_slam_amd import helper_functions as hf
    r = np.random.default_rng(0)
    for _ in range(300):
        v = r.normal(size=3) * r.choice([1e-9, 1e-3, 0.5, 3.1])
        Rm = Rotation.from_rotvec(v).as_matrix()
        T = hf.transformMatrix(v.reshape(3, 1), [1, 2, 3])
        assert isinstance(T, np.matrix) and np.abs(np.asarray(T)[:3, :3] - Rm).max() < 1e-14
        assert np.asarray(T)[:3, 3].tolist() == [1, 2, 3] and np.asarray(T)[3].tolist() == [0, 0, 0, 1]
        rv = hf.Rtorvec(Rm)
        assert rv.shape == (3, 1) and np.abs(Rotation.from_rotvec(rv.ravel()).as_matrix() - Rm).max() < 1e-13


def test_isometry_helpers():
    R = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    T = Isometry3d(R, np.array([1.0, 2, 3]))
    assert np.allclose((T * T.inverse()).matrix(), np.eye(4))
    assert np.allclose(T.matrix()[:3, 3], [1, 2, 3]) and T.orientation() is R


def _scene_map(w):
    m = Map()
    frames = []
    for i, pose in enumerate(w["poses"]):
        f = _frame(i, pose, key=(i == 0))
        if i == 0:
            m.AddFrame(0, f)
        else:
            rel = np.linalg.inv(w["poses"][i - 1]) @ pose
            m.AddParentAndPose(parent_id=i - 1, frame_id=i, frame_obj=f, rel_pose_trans=rel, pose=pose)
        frames.append(f)
    for j, X in enumerate(w["points"]):
        m.AddPoint3D(j + 1, Point(X.copy(), j + 1))
    for c, p, uv in zip(w["obs_pose"], w["obs_point"], w["obs_uv"]):
        m.GetPoint(int(p) + 1).AddFrame(frames[c], uv, np.zeros(32, np.uint8))
    return m


def test_local_ba_builds_the_same_problem_as_the_flat_arrays(oracle):
    w = ba_workload(n_cams=4, n_points=40, seed=8)
    m = _scene_map(w)
    ba = BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve)
    ba.localBundleAdjustement(m)
    meas = [np.linalg.norm((np.linalg.inv(w["poses"][i - 1]) @ w["poses"][i])[:3, 3]) for i in range(1, 4)]
    ref = oracle.ba_solve(w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"],
                          w["obs_uv"], w["K"], scale_edges=([0, 1, 2], [1, 2, 3], meas))
    for i in range(4):
        assert np.allclose(m.GetFrame(i).GetPose(), ref["poses"][i], atol=1e-12)
    assert np.allclose(m.GetPoint(7).Get3dPoint(), ref["points"][6], atol=1e-12)
    assert ba.result["iterations"] == ref["iterations"] and ba.dropped_edges == 0
    assert np.allclose(ba.get_pose(2).matrix(), ref["poses"][2]) and ba.get_point(1).shape == (3,)


def test_local_ba_scale_normalisation(oracle):
    w = ba_workload(n_cams=2, n_points=60, seed=12)
    m = _scene_map(w)
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).localBundleAdjustement(m, scale=True)
    norms = np.linalg.norm(m.GetAll3DPoints(), axis=1)
    assert np.isclose(np.median(norms), 1.0, atol=1e-9)  # LocalBA.py:178-190: everything divided by the median norm


def test_motion_only_ba_fixes_keyframes_and_points(oracle):
    w = ba_workload(n_cams=4, n_points=50, seed=14, point_sigma=0)
    m = _scene_map(w)
    m.GetFrame(2).SetAsKeyFrame()
    before = {i: m.GetFrame(i).GetPose().copy() for i in range(4)}
    pts_before = m.GetAll3DPoints().copy()
    ba = BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve)
    ba.motionOnlyBundleAdjustement(m)
    assert np.allclose(m.GetFrame(0).GetPose(), before[0], atol=1e-14) and np.allclose(m.GetFrame(2).GetPose(), before[2], atol=1e-14)
    assert not np.allclose(m.GetFrame(1).GetPose(), before[1], atol=1e-9)
    assert np.array_equal(m.GetAll3DPoints(), pts_before)
    assert ba.result["chi2_final"] < ba.result["chi2_initial"]


def test_ba_api_edge_cases(capsys):
    ba = BundleAdjustment(Camera(*ICL_NUIM_K), solver=lambda *a, **k: None)
    ba.add_point(1, np.zeros(3))
    ba.add_point(1, np.ones(3))
    assert "already existing point" in capsys.readouterr().out
    ba.add_edge(point_id=1, pose_id=5, measurement=np.zeros(2), edge_id=0)  # missing pose: dropped silently as in g2o
    assert ba.dropped_edges == 1
    with pytest.raises(NotImplementedError):
        ba.add_edge_between_poses(0, 1, np.eye(4))
    ba.optimize()  # no poses: nothing to do
    assert ba.result is None


# ---------------------------------------------------------------------------------------------- SoA mirror of the map
def _tracking_map(w, seed=0):
    """A local map built through the calls main.py makes: points created with their first observation attached, then
    per-frame AddParentAndPose + AddPointToFrameCorrespondences batches."""
    rng = np.random.default_rng(seed)
    m = Map()
    frames = [_frame(i, w["poses"][i], key=(i == 0)) for i in range(len(w["poses"]))]
    m.AddFrame(0, frames[0])
    per_frame = {i: [] for i in range(len(frames))}
    for c, p, uv in zip(w["obs_pose"], w["obs_point"], w["obs_uv"]):
        per_frame[int(c)].append((int(p) + 1, uv.astype(np.float32)))
    for pid, uv in per_frame[0]:
        pt = Point(w["points"][pid - 1].copy(), pid)
        pt.AddFrame(frames[0], uv, rng.integers(0, 256, 32, dtype=np.uint8))
        m.AddPoint3D(pid, pt)
    for i in range(1, len(frames)):
        m.AddParentAndPose(parent_id=i - 1, frame_id=i, frame_obj=frames[i], rel_pose_trans=np.eye(4), pose=w["poses"][i])
        obs = [(pid, uv) for pid, uv in per_frame[i] if pid in m.points_3d]
        m.AddPointToFrameCorrespondences([o[0] for o in obs], np.array([o[1] for o in obs]),
                                         rng.integers(0, 256, (len(obs), 32), dtype=np.uint8), frames[i])
    return m


def _poses(m):
    return np.stack([m.GetFrame(i).GetPose() for i in m.frames])


def test_soa_paths_equal_the_reference_double_loop(oracle):
    w = ba_workload(n_cams=5, n_points=70, seed=41, visibility=0.7)
    for method, kw in (("motionOnlyBundleAdjustement", {}), ("localBundleAdjustement", {}),
                       ("localBundleAdjustement", {"scale": True})):
        a, b = _tracking_map(w), _tracking_map(w)
        fast = BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve)
        slow = RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve)
        getattr(fast, method)(a, **kw)
        getattr(slow, method)(b, **kw)
        assert isinstance(fast._obs_pose, np.ndarray) and isinstance(slow._obs_pose, list)  # both paths really ran
        assert np.array_equal(np.asarray(fast._obs_pose), np.asarray(slow._obs_pose))
        assert np.array_equal(np.asarray(fast._obs_point), np.asarray(slow._obs_point))
        assert np.array_equal(np.asarray(fast._obs_uv), np.stack(slow._obs_uv))
        assert np.array_equal(_poses(a), _poses(b)) and np.array_equal(a.GetAll3DPoints(), b.GetAll3DPoints())
        assert fast.get_point(3).tolist() == slow.get_point(3).tolist()


def test_soa_image_points_query_equals_object_walk():
    w = ba_workload(n_cams=4, n_points=50, seed=43, visibility=0.6)
    m = _tracking_map(w)
    for fid in (0, 2, 3, 9):
        uv, desc, xyz, ids = m.GetImagePointsWithFrameID(fid)
        m2 = _tracking_map(w)
        m2._soa.n_obs = -1  # break the mirror: the object walk answers, after a rebuild without usable descriptors?
        ruv, rdesc, rxyz, rids = [], [], [], []
        for p in m2.points_3d.values():
            hit = p.frames.get(fid)
            if hit is not None:
                ruv.append(hit[1]); rdesc.append(hit[2]); rxyz.append(p.location_3d); rids.append(p.ID)
        assert np.array_equal(ids, np.array(rids)) and np.array_equal(np.asarray(uv), np.array(ruv).reshape(-1, 2) if ruv else np.array(ruv))
        if ruv:
            assert np.array_equal(desc, np.array(rdesc)) and np.array_equal(xyz, np.array(rxyz)) and uv.dtype == np.float32


def test_soa_survives_edits_behind_the_maps_back(oracle):
    w = ba_workload(n_cams=4, n_points=40, seed=45, visibility=0.8)
    a, b = _tracking_map(w), _tracking_map(w)
    for m in (a, b):
        m.soa()                                                  # mirror built and valid
        p = m.GetPoint(5)
        p.AddFrame(m.GetFrame(3), np.array([123.0, 45.0], np.float32), np.zeros(32, np.uint8))  # direct edit (re-observation or new)
        m.GetPoint(7).UpdatePoint(np.array([0.1, 0.2, 4.0]))      # xyz rebinding
        m.UpdatePoint3D(np.array([0.3, -0.2, 3.5]), 9)
        m.DiscardOutlierMapPoints(n_visible_frames=2)            # new dict object
        extra = Point(np.array([0.0, 0.0, 5.0]), 999)
        extra.AddFrame(m.GetFrame(1), np.array([300.0, 200.0], np.float32), np.zeros(32, np.uint8))
        m.Store3DPoints({999: extra})
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(a)
    RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(b)
    assert np.array_equal(_poses(a), _poses(b))
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).localBundleAdjustement(a)
    RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).localBundleAdjustement(b)
    assert np.array_equal(_poses(a), _poses(b)) and np.array_equal(a.GetAll3DPoints(), b.GetAll3DPoints())


def test_local_map_copy_uses_only_frames_present_in_the_map(oracle):
    """main.py:333-345: the local map holds copies of the points with only the key frame's observation."""
    w = ba_workload(n_cams=4, n_points=40, seed=47)
    g = _tracking_map(w)
    local_a, local_b = Map(), Map()
    for lm in (local_a, local_b):
        kf = g.GetFrame(0)
        lm.AddFrame(0, kf)
        lm.Store3DPoints(g.GetCopyOfPointObjects(0))
        f = _frame(1, w["poses"][1])
        lm.AddParentAndPose(parent_id=0, frame_id=1, frame_obj=f, rel_pose_trans=np.eye(4), pose=w["poses"][1])
        uv, desc, xyz, ids = lm.GetImagePointsWithFrameID(0)
        lm.AddPointToFrameCorrespondences(ids[::2], uv[::2] + 1.5, desc[::2], f)
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(local_a)
    RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(local_b)
    assert np.array_equal(_poses(local_a), _poses(local_b))
    assert not np.array_equal(local_a.GetFrame(1).GetPose(), w["poses"][1])


def test_soa_notices_an_overwritten_observation(oracle):
    w = ba_workload(n_cams=3, n_points=30, seed=49)
    a, b = _tracking_map(w), _tracking_map(w)
    for m in (a, b):
        m.soa()
        m.GetPoint(4).AddFrame(m.GetFrame(2), np.array([50.0, 60.0], np.float32), np.zeros(32, np.uint8))  # same count
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(a)
    RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(b)
    assert np.array_equal(_poses(a), _poses(b))


def test_correspondences_from_a_generator_and_the_read_only_cached_answer():
    """Round-2 advisor: AddPointToFrameCorrespondences walks point_ids more than once (mirror, then the Point objects),
    so a generator -- which the reference's zip() accepts -- must be materialised first; and the cached answer of
    GetImagePointsWithFrameID is handed to every caller as the same arrays, so they are read-only."""
    m = Map()
    f0, f1 = _frame(0, np.eye(4), key=True), _frame(1, np.eye(4))
    m.AddFrame(0, f0)
    m.AddParentAndPose(parent_id=0, frame_id=1, frame_obj=f1, rel_pose_trans=np.eye(4), pose=np.eye(4))
    for pid in range(1, 7):
        p = Point(np.array([pid, 0.0, 5.0]), pid)
        p.AddFrame(f0, np.array([10.0 * pid, 20.0]), np.full(32, pid, np.uint8))
        m.AddPoint3D(pid, p)
    m.GetImagePointsWithFrameID(0)  # the mirror is in sync: the batch below takes the array fast path
    uv1 = np.arange(8.0).reshape(4, 2)
    m.AddPointToFrameCorrespondences((pid for pid in (1, 2, 4, 6)), uv1, np.zeros((4, 32), np.uint8), f1)
    assert [pid for pid in range(1, 7) if m.GetPoint(pid).IsVisibleTo(1)] == [1, 2, 4, 6]       # the Point objects ...
    assert [m.GetPoint(pid).GetNVisibleFrames() for pid in (1, 3)] == [2, 1]
    uv, desc, xyz, ids = m.GetImagePointsWithFrameID(1)                                          # ... and the mirror
    assert ids.tolist() == [1, 2, 4, 6] and np.array_equal(uv, uv1)
    m.DiscardOutlierMapPoints(n_visible_frames=2)
    assert sorted(m.points_3d) == [1, 2, 4, 6]
    a = m.GetImagePointsWithFrameID(1)
    b = m.GetImagePointsWithFrameID(1)
    if a[0] is b[0]:  # served from the cache: the very same objects, hence read-only
        for arr in a:
            with pytest.raises(ValueError):
                arr[...] = 0
    assert np.array_equal(b[0], uv1) and b[3].tolist() == [1, 2, 4, 6]


def _small_map(n=40, seed=3):
    rng = np.random.default_rng(seed)
    m = Map()
    f0, f1 = _frame(0, np.eye(4), key=True), _frame(1, np.eye(4), key=True)
    m.AddFrame(0, f0)
    m.AddFrame(1, f1)
    for pid in range(1, n + 1):
        p = Point(rng.normal(size=3), pid)
        p.AddFrame(f0, rng.uniform(0, 600, 2).astype(np.float32), rng.integers(0, 256, 32).astype(np.uint8))
        m.AddPoint3D(pid, p)
    m.AddPointToFrameCorrespondences(list(range(1, n + 1, 2)), rng.uniform(0, 600, (n // 2, 2)).astype(np.float32),
                                     rng.integers(0, 256, (n // 2, 32)).astype(np.uint8), f1)
    return m, rng


def test_bulk_point_write_back_equals_one_update_per_point():
    """Map._update_points (the BA write-back of all points at once) leaves the map as P calls of UpdatePoint3D would
    (LocalBA.py:189-190) -- every answer the map gives, the mirror, the rebinding of location_3d -- and a second map that holds
    the same Point objects hears of the change."""
    a, rng = _small_map()
    b, _ = _small_map()
    other = Map()                                    # shares a's Point objects (a local map made without copying)
    other.AddFrame(0, a.GetFrame(0))
    for pid, p in list(a.points_3d.items())[:10]:
        other.AddPoint3D(pid, p)
    assert np.array_equal(other.GetImagePointsWithFrameID(0)[2], a.GetImagePointsWithFrameID(0)[2][:10])
    new = rng.normal(size=(40, 3))
    a.GetImagePointsWithFrameID(0)                   # (answers cached before the write-back must not survive it)
    a._update_points(new)
    for i, pid in enumerate(b.points_3d):
        b.UpdatePoint3D(new[i], pid)
    for fid in (0, 1):
        for x, y in zip(a.GetImagePointsWithFrameID(fid), b.GetImagePointsWithFrameID(fid)):
            assert np.array_equal(x, y)
    assert np.array_equal(a.GetAll3DPoints(), new) and np.array_equal(a.soa().xyz[:40], new)
    assert all(p.location_3d is r for p, r in zip(a.points_3d.values(), a.soa().xyz_refs))
    assert np.array_equal(other.GetImagePointsWithFrameID(0)[2], new[:10])       # the sharing map follows
    # the mirror is in sync afterwards: soa() does not rebuild (same object, same generation) and later edits still register
    s = a.soa()
    a.points_3d[3].UpdatePoint(np.array([9.0, 9.0, 9.0]))
    assert a.soa() is s and np.array_equal(a.GetImagePointsWithFrameID(0)[2][2], [9.0, 9.0, 9.0])
    # a mask (LocalBA.py:147-151 branch) and a wrong shape
    keep = np.zeros(40, bool)
    keep[::4] = True
    newer = rng.normal(size=(40, 3))
    a._update_points(newer, keep)
    got = a.GetAll3DPoints()
    assert np.array_equal(got[::4], newer[::4]) and np.array_equal(got[1], new[1])
    with pytest.raises(ValueError):
        a._update_points(new[:5])


def test_local_ba_write_back_goes_through_the_bulk_path(oracle):
    w = ba_workload(n_cams=3, n_points=30, seed=11)
    m = _scene_map(w)
    gen = m.soa().gen
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).localBundleAdjustement(m)
    assert m.soa().gen == gen                                   # the mirror was not rebuilt by the write-back ...
    X = m.GetAll3DPoints()
    assert np.array_equal(m.soa().xyz[:len(X)], X)             # ... and holds what the Point objects hold
    assert not np.allclose(X, w["points"])                    # (the solve moved the points)


def test_get_list_diff_on_float32_key_points_keeps_the_loop_semantics():
    """helper_functions.GetListDiff (helper_functions.py:316-324's loop: both coordinates equal): the float32 fast path against
    the loop itself, including a negative zero (equal to +0.0) and a NaN (equal to nothing)."""
    from visual_slam_amd import helper_functions as hf
    rng = np.random.default_rng(5)
    kp1 = rng.integers(0, 40, (300, 2)).astype(np.float32)
    kp2 = kp1[rng.permutation(300)[:120]].copy()
    kp1[7] = [-0.0, 5.0]
    kp2[0] = [0.0, 5.0]
    kp1[9] = [np.nan, 1.0]
    kp2[1] = [np.nan, 1.0]

    def loop(a, b):
        out = []
        for i, x in enumerate(a):
            if not any(x[0] == k[0] and x[1] == k[1] for k in b):
                out.append(i)
        return out
    want = loop(kp1, kp2)
    assert hf.GetListDiff(kp1, kp2) == want and 7 not in want and 9 in want
    assert hf.GetListDiff(kp1.astype(np.float64), kp2) == want          # mixed / other dtypes: the general path
    assert hf.GetListDiff(kp1, kp2[:0]) == list(range(300)) and hf.GetListDiff(kp1[:0], kp2) == []


def test_lazy_local_map_copies_equal_the_eager_ones(oracle):
    """Map.GetCopyOfPointObjects hands out a dict whose Point copies are made when somebody looks at them, and Map.Store3DPoints
    seeds the local map's mirror from the same arrays (main.py:345).  Every observable must equal the object-walk version's:
    keys and their order, ids, positions (own copies), the one observation each, and everything a local map does afterwards."""
    def local_maps(lazy):
        Map.use_lazy_copies = lazy
        try:
            g, rng = _small_map(60, seed=9)
            copies = g.GetCopyOfPointObjects(1)
            lm = Map()
            key = g.GetFrame(1)
            lm.AddFrame(1, key)
            lm.Store3DPoints(copies)
            return g, lm, copies, rng
        finally:
            Map.use_lazy_copies = True
    g0, a, ca, rng = local_maps(False)
    g1, b, cb, _ = local_maps(True)
    assert type(ca) is dict and type(cb).__name__ == "_LazyPoints" and len(ca) == len(cb) == 30
    # the tracking loop's view of the local map, before any Point of the lazy one exists
    for x, y in zip(a.GetImagePointsWithFrameID(1), b.GetImagePointsWithFrameID(1)):
        assert np.array_equal(x, y)
    f2a, f2b = _frame(2, np.eye(4)), _frame(2, np.eye(4))
    ids = list(range(1, 61, 2))[::3]
    uv2 = rng.uniform(0, 600, (len(ids), 2)).astype(np.float32)
    d2 = rng.integers(0, 256, (len(ids), 32)).astype(np.uint8)
    for m, f in ((a, f2a), (b, f2b)):
        m.AddParentAndPose(parent_id=1, frame_id=2, frame_obj=f, rel_pose_trans=np.eye(4), pose=np.eye(4))
        m.AddPointToFrameCorrespondences(ids, uv2, d2, f)
    assert cb._seed is not None                                   # still nothing materialised
    for fid in (1, 2):
        for x, y in zip(a.GetImagePointsWithFrameID(fid), b.GetImagePointsWithFrameID(fid)):
            assert np.array_equal(x, y)
    # the motion-only problem built from either local map is the same
    pa, pb = BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve), BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve)
    pa._graph_from_soa(a, lambda i, f: f.IsKeyFrame(), True, False)
    pb._graph_from_soa(b, lambda i, f: f.IsKeyFrame(), True, False)
    for k in ("_obs_pose", "_obs_point", "_obs_uv", "_points", "_point_fixed"):
        assert np.array_equal(getattr(pa, k), getattr(pb, k)), k
    assert pa._point_ids == pb._point_ids and list(pa._point_ids) == list(pb._point_ids)
    # now look at the objects
    assert list(a.points_3d.keys()) == list(b.points_3d.keys()) and cb._seed is None
    for (ka, p), (kb, q) in zip(a.points_3d.items(), b.points_3d.items()):
        orig = g1.points_3d[kb]
        assert ka == kb == p.ID == q.ID and np.array_equal(p.location_3d, q.location_3d) and q.location_3d is not orig.location_3d
        assert list(p.frames) == list(q.frames) and q._rev == p._rev
        for fid in p.frames:
            (fa, ua, da), (fb, ub, db) = p.frames[fid], q.frames[fid]
            assert np.array_equal(ua, ub) and np.array_equal(da, db) and fa.ID == fb.ID
        assert q.frames[1][0] is g1.GetFrame(1) and q.IsVisibleTo(1) and q.IsVisibleTo(2) == p.IsVisibleTo(2)
    # edits of a materialised copy are the local map's business only, and are noticed by its mirror
    q = b.points_3d[ids[0]]
    q.UpdatePoint(np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(b.GetImagePointsWithFrameID(1)[2][0], [1.0, 2.0, 3.0])
    assert not np.array_equal(g1.points_3d[ids[0]].location_3d, [1.0, 2.0, 3.0])
    # dict protocol of the lazy object
    g2, _, _, _ = local_maps(True)
    c = g2.GetCopyOfPointObjects(1)
    assert len(c) == 30 and c._seed is not None and (3 in c) and c._seed is None and 2 not in c
    assert {**g2.GetCopyOfPointObjects(1)}.keys() == ca.keys() and dict(g2.GetCopyOfPointObjects(1)).keys() == ca.keys()
    assert [k for k in g2.GetCopyOfPointObjects(1)] == list(ca)
    # the mirror cannot vouch for the answer -> the object walk: a frame re-numbered after its observations were recorded
    g3, _ = _small_map(20, seed=2)
    g3.GetFrame(1).AddID(7)
    assert type(g3.GetCopyOfPointObjects(1)) is dict and len(g3.GetCopyOfPointObjects(1)) == 0
    with pytest.raises(KeyError):
        g3.GetCopyOfPointObjects(7)          # visible to "7" by the Frame's id, recorded under key 1 (map.py:67, as the reference)


def test_rows_pack_helper():
    """visual_slam_amd._rows.pack: a list of equally long contiguous rows -> the rows of one array; anything else is refused (the
    caller then converts with numpy)."""
    from visual_slam_amd import _rows
    from visual_slam_amd.map import _rows_of
    rng = np.random.default_rng(0)
    base = rng.normal(size=(50, 3))
    out = np.empty((50, 3))
    assert _rows.pack(list(base), out) is True and np.array_equal(out, base)
    assert _rows.pack(list(base[::-1]), out) is True and np.array_equal(out, base[::-1])            # any order, any owner
    assert _rows.pack(list(base), np.empty((50, 3), np.float32)) is False                           # another element type
    assert _rows.pack(list(base[:49]) + [base[0, :2]], out) is False                                # a ragged row
    assert _rows.pack([base[:, 0]] * 3, np.empty((3, 50))) is False                                 # strided rows
    assert _rows.pack([[1.0, 2.0, 3.0]] * 2, np.empty((2, 3))) is False                             # not buffers
    with pytest.raises(TypeError):
        _rows.pack(list(base), np.zeros((50, 3)).tobytes())                                        # read-only output
    d = rng.integers(0, 256, (20, 32)).astype(np.uint8)
    assert np.array_equal(_rows_of(list(d)), d) and _rows_of(list(d)).dtype == np.uint8
    assert np.array_equal(_rows_of(list(base), np.float64), base)
    assert np.array_equal(_rows_of([[1.0, 2.0], [3.0, 4.0]], np.float64), [[1.0, 2.0], [3.0, 4.0]])   # plain lists: numpy's way
    assert _rows_of([np.zeros(3), np.zeros(2)]) is None
    assert np.array_equal(_rows_of(list(base.astype(np.float32)), np.float64), base.astype(np.float32).astype(np.float64))


def test_mirror_remembers_the_frame_objects_of_its_observations():
    """GetCopyOfPointObjects answers from the mirror only when it knows the Frame object behind every frame id (IsVisibleTo asks
    the objects): recorded by AddPointToFrameCorrespondences, by the absorption of added points and by a rebuild from the objects;
    two different Frame objects under one id, or observations that arrived without their object, switch the fast path off."""
    m, _ = _small_map(12, seed=4)
    s = m.soa()
    assert set(s.frame_objs) == {0, 1} and s.frame_objs[0] is m.GetFrame(0) and s.frame_objs[1] is m.GetFrame(1)
    assert type(m.GetCopyOfPointObjects(1)).__name__ == "_LazyPoints"
    m.points_3d[2].AddFrame(_frame(1, np.eye(4)), np.zeros(2, np.float32), np.zeros(32, np.uint8))   # ANOTHER object with id 1
    s = m.soa()                                                                                       # (rebuilt from the objects)
    assert s.frame_objs[1] is None and type(m.GetCopyOfPointObjects(1)) is dict
    from visual_slam_amd.map import mirror_of_points, _UNKNOWN_FRAMES
    s2 = mirror_of_points(m.points_3d)
    s2.add_obs([0], 5, np.zeros((1, 2)), None)                                                       # no Frame object given
    assert s2.frame_objs is _UNKNOWN_FRAMES


def _points_equal(a, b):
    assert list(a.points_3d) == list(b.points_3d)
    for pa, pb in zip(a.points_3d.values(), b.points_3d.values()):
        assert pa.ID == pb.ID and pa._rev == pb._rev and np.array_equal(pa.location_3d, pb.location_3d)
        assert len(pa._cells) == len(pb._cells) == 1
        assert list(pa.frames) == list(pb.frames)
        for k in pa.frames:
            fa, fb = pa.frames[k], pb.frames[k]
            assert fa[0].ID == fb[0].ID and np.array_equal(fa[1], fb[1]) and np.array_equal(fa[2], fb[2])
            assert fa[1].dtype == fb[1].dtype and fa[2].dtype == fb[2].dtype


@pytest.mark.parametrize("n_obs", [1, 2, 3])
def test_points_added_as_one_batch_leave_the_map_as_the_loop_does(n_obs):
    """Map.AddPoints3D (not in the reference; the resident driver's form of main.py:130-135 / 312-318) against the loop of
    Point(...) / AddFrame / AddPoint3D it replaces: the same objects, dict order and change counters, the same rows in the same
    order in the mirror, the same answers afterwards -- also when points and observations follow, and when the mirror was
    out of sync at the time of the call."""
    from visual_slam_amd.point import Point

    def build(bulk, stale_mirror):
        r = np.random.default_rng(5)
        m = Map()
        fr = [Frame(np.zeros((4, 4, 3), np.uint8), None, i) for i in range(3)]
        for i, f in enumerate(fr):
            m.AddFrame(i, f)
        for rnd, n in enumerate((40, 25)):
            X = r.normal(size=(n, 3))
            obs = [(fr[j], r.normal(size=(n, 2)).astype(np.float32), r.integers(0, 255, (n, 32)).astype(np.uint8)) for j in range(n_obs)]
            ids = range(1 + 100 * rnd, 1 + 100 * rnd + n)
            if rnd == 1 and stale_mirror:
                m.soa()
                m.points_3d[3].AddFrame(frame=fr[2], uv=np.zeros(2, np.float32), descriptor=np.zeros(32, np.uint8))  # behind the map's back
            if bulk:
                m.AddPoints3D(ids, X, obs)
            else:
                for k, pid in enumerate(ids):
                    p = Point(location=X[k], id=pid)
                    for f, uv, d in obs:
                        p.AddFrame(frame=f, uv=uv[k], descriptor=d[k])
                    m.AddPoint3D(pid, p)
            if rnd == 0:
                m.GetImagePointsWithFrameID(0)   # the mirror is looked at between the two rounds
        # observations of a later frame for some of the points, through the map
        sel = list(m.points_3d)[::3]
        m.AddPointToFrameCorrespondences(point_ids=sel, image_points=r.normal(size=(len(sel), 2)).astype(np.float32),
                                         descriptors=r.integers(0, 255, (len(sel), 32)).astype(np.uint8), frame_obj=fr[2])
        return m

    for stale in (False, True):
        a, b = build(False, stale), build(True, stale)
        _points_equal(a, b)
        assert a._cell[:2] == b._cell[:2]
        sa, sb = a.soa(), b.soa()
        for x, y in zip(sa.arrays(), sb.arrays()):
            assert np.array_equal(x, y) and x.dtype == y.dtype
        assert sa.n_points == sb.n_points and sa.rev == sb.rev and sa.fid_rows == sb.fid_rows and sa.point_slot == sb.point_slot
        assert np.array_equal(sa.xyz[:sa.n_points], sb.xyz[:sb.n_points])
        assert all(p.location_3d is r for p, r in zip(b.points_3d.values(), sb.xyz_refs))
        for f in range(3):
            for x, y in zip(a.GetImagePointsWithFrameID(f), b.GetImagePointsWithFrameID(f)):
                assert np.array_equal(x, y)
            la, lb = a.GetCopyOfPointObjects(f), b.GetCopyOfPointObjects(f)
            assert list(la.keys()) == list(lb.keys())
    # errors: a duplicate id raises with the reference's text and leaves the map untouched; ragged input
    m = build(True, False)
    n0 = len(m.points_3d)
    with pytest.raises(Exception, match="Duplicate point3d warning"):
        m.AddPoints3D([900, 1], np.zeros((2, 3)), [(m.GetFrame(0), np.zeros((2, 2)), np.zeros((2, 32), np.uint8))])
    with pytest.raises(Exception, match="Duplicate point3d warning"):
        m.AddPoints3D([900, 900], np.zeros((2, 3)), [(m.GetFrame(0), np.zeros((2, 2)), np.zeros((2, 32), np.uint8))])
    with pytest.raises(ValueError):
        m.AddPoints3D([900, 901], np.zeros((2, 3)), [(m.GetFrame(0), np.zeros((3, 2)), np.zeros((2, 32), np.uint8))])
    assert len(m.points_3d) == n0
    m.AddPoints3D([], np.zeros((0, 3)), [])
    assert len(m.points_3d) == n0
    # the mirror owns its rows: editing the arrays handed in afterwards does not reach it
    f = Frame(np.zeros((4, 4, 3), np.uint8), None, 0)
    m = Map()
    m.AddFrame(0, f)
    uv, d, X = np.ones((5, 2), np.float32), np.ones((5, 32), np.uint8), np.zeros((5, 3))
    m.AddPoints3D(range(1, 6), X, [(f, uv, d)])
    uv[:], d[:], X[:] = 7, 9, 3
    arr = m.soa().arrays()
    assert arr[2].max() == 1 and arr[3].max() == 1 and m.soa().xyz[:5].max() == 0


def test_collect_reads_added_points_in_one_pass_or_declines():
    """_rows.collect (the C pass of Map._absorb_added): positions, counter sum, observation counts and the observations point by
    point in dict order; None -- never an exception -- for objects that are not Points, wherever in the batch they stand."""
    _rows = pytest.importorskip("visual_slam_amd._rows")
    from visual_slam_amd.point import Point
    fr = [Frame(np.zeros((4, 4, 3), np.uint8), None, i) for i in range(3)]
    pts = []
    for k in range(5):
        p = Point(location=np.full(3, float(k)), id=k)
        for j in range(k % 3 + 1):
            p.AddFrame(frame=fr[j], uv=np.array([k, j], np.float32), descriptor=np.full(32, 10 * k + j, np.uint8))
        pts.append(p)
    locs, rev_sum, counts, fids, fobjs, uvs, descs = _rows.collect(pts)
    assert all(a is p._loc for a, p in zip(locs, pts)) and rev_sum == sum(p._rev for p in pts)
    assert counts == [len(p._frames) for p in pts]
    flat = [(f, t) for p in pts for f, t in p._frames.items()]
    assert fids == [f for f, _ in flat] and all(a is t[0] for a, (_, t) in zip(fobjs, flat))
    assert all(a is t[1] for a, (_, t) in zip(uvs, flat)) and all(a is t[2] for a, (_, t) in zip(descs, flat))
    assert _rows.collect([]) == ([], 0, [], [], [], [], [])
    assert _rows.collect(tuple(pts))[2] == counts

    class NotAPoint:
        pass

    class HalfAPoint:
        _loc, _rev, _frames = np.zeros(3), 0, {0: "not a triple"}

    for bad in (NotAPoint(), HalfAPoint(), 7):
        for where in (0, 2, 5):
            seq = list(pts)
            seq.insert(where, bad)
            assert _rows.collect(seq) is None
    # ... and a map that meets such an object among its added points rebuilds its mirror from the objects instead
    m = Map()
    m.AddFrame(0, fr[0])
    for k, p in enumerate(pts):
        m.AddPoint3D(k, p)
    assert m.soa().n_points == 5 and m.soa().n_obs == sum(counts)
