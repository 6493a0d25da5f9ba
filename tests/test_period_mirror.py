"""CPU: the routing of BundleAdjustment.motionOnlyBundleAdjustement to a device-resident tracking period behind Map
(map.py::_PeriodMirror) -- incremental pushes, and every way the period must be restarted or declined.
The GPU context is replaced by a stand-in with the same four methods that keeps the period as arrays and solves it with the
CPU oracle, so the logic (not the kernels) is covered without a GPU; tests/test_gpu_api.py covers the real thing."""
import numpy as np
import pytest

from visual_slam_amd.LocalBA import BundleAdjustment, Camera
from visual_slam_amd.frame import Frame
from visual_slam_amd.map import Map
from visual_slam_amd.point import Point
from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload
from oracle.ref_graph import RefLoopBundleAdjustment


class FakePeriodContext:
    """Context.track_begin / track_push_frame / track_end with the oracle as the solver."""

    def __init__(self, oracle):
        self.oracle = oracle
        self._track = None
        self._track_owner = None
        self.begins = 0
        self.general = 0
        self.pushes = []

    def track_begin(self, xyz, desc, key_pose, K, max_frames=64, max_kp=3000, pnp_iterations=100):
        self._track_owner = None
        self._track = dict(xyz=np.array(xyz, np.float64), poses=[np.array(key_pose, np.float64).reshape(4, 4)], K=K,
                           obs=[], cap=max_frames)
        self.begins += 1

    def track_push_frame(self, point_idx, uv, pose, lm_iterations=10, huber_delta=float(np.sqrt(5.991))):
        t = self._track
        assert t is not None and len(t["poses"]) - 1 < t["cap"]
        k = len(t["poses"])
        t["poses"].append(np.array(pose, np.float64).reshape(4, 4))
        t["obs"].append((np.full(len(point_idx), k, np.int32), np.asarray(point_idx, np.int32), np.asarray(uv, np.float64).reshape(-1, 2)))
        self.pushes.append((len(point_idx), lm_iterations))
        if lm_iterations > 0:
            fixed = np.zeros(len(t["poses"]), np.uint8)
            fixed[0] = 1
            r = self.oracle.ba_solve(np.stack(t["poses"]), fixed, t["xyz"], np.ones(len(t["xyz"]), np.uint8),
                                     np.concatenate([o[0] for o in t["obs"]]), np.concatenate([o[1] for o in t["obs"]]),
                                     np.concatenate([o[2] for o in t["obs"]]), t["K"], huber_delta=huber_delta,
                                     max_iterations=lm_iterations)
            t["poses"] = [p.copy() for p in r["poses"]]
        return np.stack(t["poses"])

    def track_end(self):
        self._track = None
        self._track_owner = None

    def ba_solve(self, *a, **k):  # the general path (vs_ba_solve)
        self.general += 1
        return self.oracle.ba_solve(*a, **k)


def _frame(fid, pose, key=False):
    f = Frame(np.zeros((4, 4, 3), np.uint8), None, fid)
    f.AddPose(np.array(pose))
    if key:
        f.SetAsKeyFrame()
    return f


def _period(w, n_frames, edits=None, ctx=None, use_mirror=True):
    """Builds the local map frame by frame the way main.py:181-214 does and runs the motion-only BA after every frame."""
    rng = np.random.default_rng(0)
    m = Map()
    m.use_device_mirror = use_mirror
    key = _frame(0, w["poses"][0], key=True)
    m.AddFrame(0, key)
    per_frame = {i: [] for i in range(len(w["poses"]))}
    for c, p, uv in zip(w["obs_pose"], w["obs_point"], w["obs_uv"]):
        per_frame[int(c)].append((int(p) + 1, uv.astype(np.float32)))
    for pid, uv in per_frame[0]:
        pt = Point(w["points"][pid - 1].copy(), pid)
        pt.AddFrame(key, uv, rng.integers(0, 256, 32, dtype=np.uint8))
        m.AddPoint3D(pid, pt)
    for k in range(1, n_frames):
        f = _frame(k, w["poses"][k])
        m.AddParentAndPose(parent_id=k - 1, frame_id=k, frame_obj=f, rel_pose_trans=np.eye(4), pose=w["poses"][k])
        obs = [(pid, uv) for pid, uv in per_frame[k] if pid in m.points_3d]
        m.AddPointToFrameCorrespondences([o[0] for o in obs], np.array([o[1] for o in obs]),
                                         rng.integers(0, 256, (len(obs), 32), dtype=np.uint8), f)
        if edits and k in edits:
            edits[k](m)
        if ctx is not None:
            BundleAdjustment(Camera(*ICL_NUIM_K), context=ctx).motionOnlyBundleAdjustement(m)
        else:
            RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=_period.oracle.ba_solve).motionOnlyBundleAdjustement(m)
    return m


def _poses(m):
    return np.stack([np.asarray(m.GetFrame(i).GetPose()) for i in m.frames])


@pytest.fixture()
def scene(oracle):
    _period.oracle = oracle
    return ba_workload(n_cams=6, n_points=80, seed=51, visibility=0.8, point_sigma=0)


def test_incremental_pushes_equal_the_reference_loop(scene, oracle):
    ctx = FakePeriodContext(oracle)
    a = _period(scene, 6, ctx=ctx)
    b = _period(scene, 6)
    assert ctx.begins == 1 and [it for _, it in ctx.pushes] == [10] * 5      # one period, one push per frame
    assert np.allclose(_poses(a), _poses(b), rtol=0, atol=2e-8)


def test_edits_behind_the_mirror_restart_the_period(scene, oracle):
    def move_point(m):
        m.UpdatePoint3D(np.asarray(m.GetPoint(3).Get3dPoint()) + 0.02, 3)

    def replace_pose_value(m):
        m.UpdatePose(np.asarray(m.GetFrame(1).GetPose()) @ np.array([[1, 0, 0, 0.01], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]), 1)

    def same_pose_new_object(m):
        m.UpdatePose(np.array(m.GetFrame(2).GetPose()), 2)

    def direct_observation(m):
        m.GetPoint(5).AddFrame(m.GetFrame(1), np.array([300.0, 200.0], np.float32), np.zeros(32, np.uint8))

    for edits, restarts in (({3: move_point}, 2), ({3: replace_pose_value}, 2), ({4: same_pose_new_object}, 1),
                            ({4: direct_observation}, None)):
        ctx = FakePeriodContext(oracle)
        a = _period(scene, 6, edits=edits, ctx=ctx)
        b = _period(scene, 6, edits=edits)
        assert np.allclose(_poses(a), _poses(b), rtol=0, atol=2e-8), list(edits)  # the stand-in sums chi2 frame-major: LM paths agree to ~1e-9
        if restarts is not None:
            assert ctx.begins == restarts and ctx.general == 0, (list(edits), ctx.begins)
        else:
            assert ctx.general >= 1   # observations re-attached one by one no longer look like per-frame batches
        # a restart re-sends the earlier frames without solving, then solves on the last one
        if restarts == 2:
            assert (0 in [it for _, it in ctx.pushes])


def test_maps_that_are_not_a_tracking_period_take_the_general_path(scene, oracle):
    ctx = FakePeriodContext(oracle)
    m = _period(scene, 4, ctx=ctx)
    n = len(ctx.pushes)
    m.GetFrame(2).SetAsKeyFrame()                       # a second key frame: not a period any more
    BundleAdjustment(Camera(*ICL_NUIM_K), context=ctx).motionOnlyBundleAdjustement(m)
    assert len(ctx.pushes) == n and ctx.general == 1
    # an explicitly opened period on the context is never taken over
    ctx2 = FakePeriodContext(oracle)
    ctx2.track_begin(np.zeros((1, 3)), np.zeros((1, 32), np.uint8), np.eye(4), ICL_NUIM_K)
    m2 = Map()
    assert _period(scene, 2, ctx=None) is not None
    m3 = _period(scene, 1, ctx=None)
    f = _frame(1, scene["poses"][1])
    m3.AddParentAndPose(parent_id=0, frame_id=1, frame_obj=f, rel_pose_trans=np.eye(4), pose=scene["poses"][1])
    assert m3.resident_motion_ba(ctx2, ICL_NUIM_K, 2.0, 10) is None and ctx2.begins == 1


def test_switch(scene, oracle):
    ctx = FakePeriodContext(oracle)
    Map.use_device_mirror = False
    try:
        m = _period(scene, 1, ctx=None)
        assert m.resident_motion_ba(ctx, ICL_NUIM_K, 2.0, 10) is None
    finally:
        Map.use_device_mirror = True
