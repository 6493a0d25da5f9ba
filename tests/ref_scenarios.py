"""Scripted scenarios that are run TWICE: by tests/golden/make_ref_fixtures.py against the reference's own classes and
functions (their source taken out of /root/reference with `ast` and executed in the build container -- the module is never
imported, cv2 / g2o are neither installed nor faked), and by the tests against the product's classes.  The recorded values of
the first run are committed as tests/golden/ref_fixtures.npz; the tests compare the second run with them.

Nothing here touches /root/reference: the scripts only take the classes / callables they are handed.
"""
import numpy as np


def _rng(seed):
    return np.random.default_rng(seed)


def _pose(rng, i):
    """a camera-to-world 4x4 (rotation about y by a few degrees, translation along an arc)"""
    a = 0.05 * i + 0.01 * rng.standard_normal()
    R = np.array([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = [0.1 * i, 0.01 * rng.standard_normal(), 0.02 * i]
    return T


def build_map(Map, Point, make_frame, seed=11, n_frames=5, n_points=40, key_every=2):
    """A map built through the reference's own call sequence (main.py:116-135, 208-210, 318-319): frames through
    AddFrame / AddParentAndPose, points through AddPoint3D, observations through Point.AddFrame and
    AddPointToFrameCorrespondences."""
    rng = _rng(seed)
    m = Map()
    frames = []
    for i in range(n_frames):
        f = make_frame(i)
        if i % key_every == 0:
            f.SetAsKeyFrame()
        pose = _pose(rng, i)
        if i == 0:
            f.AddPose(pose)
            m.AddFrame(0, f)
        else:
            rel = np.linalg.inv(frames[-1].GetPose()) @ pose
            m.AddParentAndPose(parent_id=i - 1, frame_id=i, frame_obj=f, rel_pose_trans=rel, pose=pose)
        frames.append(f)
    # point ids are not contiguous and not sorted on purpose (dict insertion order is what the reference walks)
    ids = [int(x) for x in rng.permutation(np.arange(100, 100 + 3 * n_points, 3))[:n_points]]
    for pid in ids:
        p = Point(rng.uniform(-2, 2, 3) + np.array([0.0, 0.0, 4.0]), pid)
        p.AddFrame(frames[0], rng.uniform(0, 640, 2).astype(np.float32), rng.integers(0, 256, 32, dtype=np.uint8))
        m.AddPoint3D(pid, p)
    for i in range(1, n_frames):
        seen = [pid for pid in ids if rng.random() < 0.6]
        uv = rng.uniform(0, 640, (len(seen), 2)).astype(np.float32)
        desc = rng.integers(0, 256, (len(seen), 32), dtype=np.uint8)
        m.AddPointToFrameCorrespondences(seen, uv, desc, frames[i])
    return m, frames, ids


def _stack(rows, shape_tail, dtype=np.float64):
    return np.asarray(rows, dtype=dtype).reshape((-1,) + tuple(shape_tail))


def map_script(Map, Point, make_frame):
    """Every getter / mutator of Map and Point the hot path uses, answers recorded as arrays."""
    out = {}
    m, frames, ids = build_map(Map, Point, make_frame)
    out["point_order"] = np.asarray([p.GetID() for p in m.points_3d.values()], np.int64)
    for fid in (0, 1, 3, 4, 9):
        uv, desc, xyz, pids = m.GetImagePointsWithFrameID(fid)
        out["img_uv_%d" % fid] = _stack(uv, (2,))
        out["img_desc_%d" % fid] = _stack(desc, (32,), np.int64)
        out["img_xyz_%d" % fid] = _stack(xyz, (3,))
        out["img_ids_%d" % fid] = np.asarray(pids, np.int64).reshape(-1)
    out["visible_0_1"] = np.asarray(m.GetPointsVisibleToFrames([0, 1]), np.int64)
    out["visible_1_2_3"] = np.asarray(m.GetPointsVisibleToFrames([1, 2, 3]), np.int64)
    out["visible_none"] = np.asarray(m.GetPointsVisibleToFrames([]), np.int64)
    out["xyz_with_ids"] = np.asarray(m.Get3DPointsWithIDs(ids[5:12]), np.float64)
    out["all_xyz"] = np.asarray(m.GetAll3DPoints(), np.float64)
    out["all_poses"] = np.stack([np.asarray(p, np.float64) for p in m.GetAllPoses()])
    out["n_visible"] = np.asarray([m.GetPoint(pid).GetNVisibleFrames() for pid in ids], np.int64)
    out["vector_norm"] = np.asarray([m.GetPoint(pid).GetVectorNorm() for pid in ids], np.float64)
    out["is_visible_2"] = np.asarray([m.GetPoint(pid).IsVisibleTo(2) for pid in ids], np.int64)
    out["get_frame_none"] = np.asarray([m.GetPoint(pid).GetFrame(7) is None for pid in ids[:4]], np.int64)
    out["parents"] = np.asarray([list(f.GetParentIDs()) + [-1] * (1 - len(list(f.GetParentIDs()))) for f in frames], np.int64)
    out["transition_3"] = np.asarray(frames[3].GetTransitionWithParentID(2), np.float64)
    out["keyframes"] = np.asarray([bool(f.IsKeyFrame()) for f in frames], np.int64)
    # local-map copy (main.py:333-336): copies carry only the key frame's observation
    cp = m.GetCopyOfPointObjects(2)
    out["copy_ids"] = np.asarray(list(cp.keys()), np.int64)
    out["copy_frames"] = np.asarray([list(p.frames.keys()) for p in cp.values()], np.int64).reshape(len(cp), -1)
    out["copy_is_deep"] = np.asarray([cp[k] is not m.GetPoint(k) for k in cp], np.int64)
    # mutators
    new_pose = np.eye(4)
    new_pose[:3, 3] = [1.0, 2.0, 3.0]
    m.UpdatePose(new_pose, 2)
    m.UpdatePoint3D(np.array([9.0, 8.0, 7.0]), ids[3])
    out["pose_after_update"] = np.asarray(m.GetFrame(2).GetPose(), np.float64)
    out["xyz_after_update"] = np.asarray(m.GetAll3DPoints(), np.float64)
    errors = []
    for fn in (lambda: m.AddFrame(1, frames[1]), lambda: m.AddPoint3D(ids[0], Point(np.zeros(3), ids[0])),
               lambda: m.UpdatePose(np.eye(4), 77), lambda: m.UpdatePoint3D(np.zeros(3), 5)):
        try:
            fn()
            errors.append("")
        except Exception as e:  # noqa: BLE001  (the reference raises bare Exception)
            errors.append(str(e))
    out["errors"] = np.asarray(errors)
    extra = Point(np.array([0.5, 0.5, 3.0]), 7)
    extra.AddFrame(frames[1], np.array([11.0, 12.0], np.float32), np.full(32, 7, np.uint8))
    m.Store3DPoints({7: extra})
    out["order_after_store"] = np.asarray(list(m.points_3d.keys()), np.int64)
    m.DiscardOutlierMapPoints(n_visible_frames=3)
    out["order_after_discard"] = np.asarray(list(m.points_3d.keys()), np.int64)
    uv, desc, xyz, pids = m.GetImagePointsWithFrameID(1)
    out["img_ids_1_after_discard"] = np.asarray(pids, np.int64).reshape(-1)
    out["img_uv_1_after_discard"] = _stack(uv, (2,))
    return out


class GraphRecorder:
    """Stands where the reference's `self` (a g2o.SparseOptimizer subclass, LocalBA.py:20) stands when its graph-building
    methods run: records every add_pose / add_point / add_edge / AddScalingEdge call with the arguments it was given.
    optimize() changes nothing; get_pose / get_point return what was put in (so the write-back code of the method runs)."""

    class _Iso:
        def __init__(self, m):
            self._m = np.array(m, np.float64)

        def matrix(self):
            return self._m.copy()

    def __init__(self):
        self.poses, self.points, self.edges, self.scale = [], [], [], []
        self._pose, self._point = {}, {}
        self.optimized = 0

    def add_pose(self, pose_id, pose, fixed=False):
        self.poses.append((pose_id, np.array(pose, np.float64), bool(fixed)))
        self._pose[pose_id] = np.array(pose, np.float64)

    def add_point(self, point_id, point, fixed=False, marginalized=True):
        self.points.append((point_id, np.array(point, np.float64), bool(fixed)))
        self._point[point_id] = np.array(point, np.float64)

    def add_edge(self, point_id, pose_id, measurement, edge_id, information=None, robust_kernel=None):
        self.edges.append((point_id, pose_id, np.array(measurement, np.float64), edge_id))

    def AddScalingEdge(self, parent_id, child_id, measurement, information=None, robust_kernel=None):
        self.scale.append((parent_id, child_id, np.array(measurement, np.float64)))

    def optimize(self, max_iterations=10, verbose=True):
        self.optimized += 1

    def get_pose(self, pose_id):
        return self._Iso(self._pose[pose_id])

    def get_point(self, point_id):
        return self._point[point_id].copy()

    def arrays(self):
        """the problem in the flat form vs_ba_solve takes: vertices in call order, edges in call order"""
        pose_idx = {pid: i for i, (pid, _, _) in enumerate(self.poses)}
        point_idx = {pid: i for i, (pid, _, _) in enumerate(self.points)}
        return {
            "pose_ids": np.asarray([p[0] for p in self.poses], np.int64),
            "poses": np.stack([p[1] for p in self.poses]) if self.poses else np.zeros((0, 4, 4)),
            "pose_fixed": np.asarray([p[2] for p in self.poses], np.int64),
            "point_ids": np.asarray([p[0] for p in self.points], np.int64),
            "points": np.stack([p[1] for p in self.points]) if self.points else np.zeros((0, 3)),
            "point_fixed": np.asarray([p[2] for p in self.points], np.int64),
            "obs_pose": np.asarray([pose_idx[e[1]] for e in self.edges], np.int64),
            "obs_point": np.asarray([point_idx[e[0]] for e in self.edges], np.int64),
            "obs_uv": np.stack([e[2] for e in self.edges]) if self.edges else np.zeros((0, 2)),
            "edge_ids": np.asarray([e[3] for e in self.edges], np.int64),
            "scale_parent": np.asarray([pose_idx[s[0]] for s in self.scale], np.int64),
            "scale_child": np.asarray([pose_idx[s[1]] for s in self.scale], np.int64),
            # EdgeSBAScale's measurement is the norm of the translation of the stored relative transform (LocalBA.py:126)
            "scale_meas": np.asarray([np.linalg.norm(s[2][:3, 3]) for s in self.scale], np.float64),
        }


GRAPH_CASES = (("localBundleAdjustement", {}), ("localBundleAdjustement", {"scale": True}),
               ("localBundleAdjustement", {"last_keyframe_id": 4}), ("motionOnlyBundleAdjustement", {}),
               ("motionOnlyBundleAdjustement", {"scale": True}))


def graph_case_name(i):
    name, kw = GRAPH_CASES[i]
    return "g%d_%s" % (i, "local" if name.startswith("local") else "motion")


def map_state(m):
    return {"poses_after": np.stack([np.asarray(f.GetPose(), np.float64) for f in m.frames.values()]),
            "points_after": np.asarray(m.GetAll3DPoints(), np.float64)}

# (n, pixel noise, seed) of the two-view scenes whose triangulation by the reference's own function is stored in
# tests/golden/ref_fixtures.npz (tri_tv<i>_*) and compared with vs_triangulate_dlt by tests/test_triangulate.py
TWO_VIEW_CASES = ((1, 0.0, 1), (500, 0.0, 2), (3000, 0.7, 3), (257, 2.0, 4), (800, 0.3, 7))
