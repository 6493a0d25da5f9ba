"""Parity against values the REFERENCE'S OWN SOURCE produced (tests/golden/ref_fixtures.npz, written in the build container by
tests/golden/make_ref_fixtures.py: pure-Python / NumPy functions and classes of the reference taken out of its files with
`ast` and executed; nothing of the reference's Python travels, only inputs and outputs).

Pinned by these tests: A6 ratio rule + gathers + output order, A7 / A8 Map and Point getters and mutators,
A14 / A15 graph construction (vertex order, fixed flags, edge order, scale edges) and write-back, 8f-3 DLT triangulation.
Still unpinned: everything whose arithmetic lives in cv2 / g2o (detector, descriptors, the k-NN search itself, the LM solve).
"""
import os

import numpy as np
import pytest

import ref_scenarios as sc
from conftest import GOLDEN
from oracle import np_reference as ref


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "ref_fixtures.npz"))


def _flip(X):
    return X * np.where(X[:, 3:] < 0, -1.0, 1.0)  # the sign of a singular vector is arbitrary


# ---------------------------------------------------------------------------------------------------------------- NumPy helpers
def test_numpy_restatements_equal_the_executed_reference(fx):
    """oracle/np_reference.py (the checker of the triangulation kernel) against the reference's own functions."""
    for pair in ("12", "13", "23"):
        P = {"1": fx["v1_P1"], "2": fx["v1_P2"], "3": fx["v1_P3"]}
        got = ref.triangulate(P[pair[0]], P[pair[1]], fx["tri_v1_%s_pts1" % pair], fx["tri_v1_%s_pts2" % pair])
        assert np.allclose(_flip(got), _flip(fx["tri_v1_%s_X" % pair]), rtol=0, atol=1e-13)
    assert np.array_equal(ref.camera_projection_matrix2(fx["tri_syn_w2c2"], fx["tri_syn_K"]), fx["tri_syn_P2"])
    assert np.array_equal(ref.make_homogeneous(fx["tri_syn_Xh"][:, :3]), fx["tri_syn_Xh"])
    got = ref.triangulate(fx["tri_syn_P1"], fx["tri_syn_P2"], fx["tri_syn_pts1"], fx["tri_syn_pts2"])
    assert np.allclose(_flip(got), _flip(fx["tri_syn_X"]), rtol=0, atol=1e-13)
    for case in range(len(sc.TWO_VIEW_CASES)):
        pre = "tri_tv%d_" % case
        got = ref.triangulate(fx[pre + "P1"], fx[pre + "P2"], fx[pre + "x1"], fx[pre + "x2"])
        assert np.allclose(_flip(got), _flip(fx[pre + "X"]), rtol=0, atol=1e-13)
    assert ref.triangulate(fx["tri_syn_P1"], fx["tri_syn_P2"], fx["tri_syn_pts1"][:0], fx["tri_syn_pts2"][:0]).shape == (0, 4)


def test_product_helpers_equal_the_executed_reference(fx):
    from visual_slam_amd import helper_functions as hf
    assert np.array_equal(hf.CameraProjectionMatrix2(fx["tri_syn_w2c2"], fx["tri_syn_K"]), fx["tri_syn_P2"])
    assert np.array_equal(hf.MakeHomogeneous(fx["tri_syn_Xh"][:, :3]), fx["tri_syn_Xh"])
    assert np.allclose(hf.CameraProjectionMatrix(fx["cpm_R"], fx["cpm_t"], fx["tri_syn_K"]), fx["cpm_out"], rtol=0, atol=1e-12)


def test_the_v1_held_points_triangulate_consistently(fx):
    """src/v1/testing.py:46-71 holds three cameras and one point seen by all three: the three pairwise triangulations of the
    reference's own code agree on it (a sanity check of the fixture, no product code involved)."""
    X = [fx["tri_v1_%s_X" % p][0] for p in ("12", "13", "23")]
    X = [x[:3] / x[3] for x in X]
    assert np.linalg.norm(X[0] - X[1]) < 0.02 * np.linalg.norm(X[0]) and np.linalg.norm(X[0] - X[2]) < 0.02 * np.linalg.norm(X[0])


def tri_case_bounds(tol):
    """(absolute bound on the unit 4-vectors, relative bound on the dehomogenised points) for a case whose systems have
    entries of unit scale (tol 1e-11) or of pixel scale (the v1 matrices reach 5e3: tol 1e-9)."""
    return tol, (1e-7 if tol > 1e-10 else 1e-9)


def tri_check(X4, want, tol):
    """X4: triangulated unit 4-vectors (sign-normalised), want: what the reference's triangulate returned."""
    atol, rel = tri_case_bounds(tol)
    if not np.allclose(X4, _flip(want), rtol=0, atol=atol):
        return False
    Xg, Xw = X4[:, :3] / X4[:, 3:], want[:, :3] / want[:, 3:]
    return bool(np.max(np.linalg.norm(Xg - Xw, axis=1) / np.linalg.norm(Xw, axis=1)) < rel)


def test_the_triangulation_check_rejects_a_relative_error_of_1e_8(fx):
    """The check the GPU test applies must bite: the reference's own output passes, the same output with its points moved by
    1e-8 relative does not (round 4's form of the assertion parsed as `assert (x < 1e-7) if .. else 1e-9` and asserted a
    constant for the unit-scale case)."""
    want = fx["tri_syn_X"]
    assert tri_check(_flip(want), want, 1e-11)
    X = want[:, :3] / want[:, 3:]
    moved = np.c_[X * (1 + 1e-8), np.ones(len(X))]
    moved = _flip(moved / np.linalg.norm(moved, axis=1, keepdims=True))
    assert not tri_check(moved, want, 1e-11)
    rel = np.max(np.linalg.norm(moved[:, :3] / moved[:, 3:] - X, axis=1) / np.linalg.norm(X, axis=1))
    assert 0.9e-8 < rel < 1.1e-8 and tri_case_bounds(1e-11)[1] < rel < tri_case_bounds(1e-9)[1]


@pytest.mark.gpu
def test_hip_triangulation_equals_the_executed_reference(vs, fx):
    """vs_triangulate_dlt (one-sided Jacobi SVD in registers) against what the reference's triangulate returned.
    Tolerance as tests/test_triangulate.py: 1e-11 on the unit vectors, 1e-9 relative on the points."""
    cases = [(fx["tri_syn_P1"], fx["tri_syn_P2"], fx["tri_syn_pts1"], fx["tri_syn_pts2"], fx["tri_syn_X"], 1e-11)]
    for pair in ("12", "13", "23"):
        P = {"1": fx["v1_P1"], "2": fx["v1_P2"], "3": fx["v1_P3"]}
        # entries of the v1 matrices reach 5e3 (pixels): the 4x4 systems are worse conditioned than unit-scale ones
        cases.append((P[pair[0]], P[pair[1]], fx["tri_v1_%s_pts1" % pair], fx["tri_v1_%s_pts2" % pair], fx["tri_v1_%s_X" % pair], 1e-9))
    eye = np.eye(4)
    for P1, P2, x1, x2, want, tol in cases:
        X4, _ = vs.triangulate_dlt(P1, P2, ref.make_homogeneous(x1), ref.make_homogeneous(x2), eye, eye)
        assert tri_check(X4, want, tol)
        X = X4[:, :3] / X4[:, 3:]
        moved = np.c_[X * (1 + 1e-8), np.ones(len(X))]
        assert not tri_check(_flip(moved / np.linalg.norm(moved, axis=1, keepdims=True)), want, 1e-11)


# ------------------------------------------------------------------------------------------------------------- Map / Point / Frame
def _product_frame(i):
    from visual_slam_amd.frame import Frame
    return Frame(np.zeros((4, 4, 3), np.uint8), None, i)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in "US":
        return a.tolist() == b.tolist()
    return a.shape == b.shape and np.array_equal(a, b)


def test_map_and_point_behave_as_the_executed_reference(fx):
    from visual_slam_amd.map import Map
    from visual_slam_amd.point import Point
    got = sc.map_script(Map, Point, _product_frame)
    want = {k[4:]: fx[k] for k in fx.files if k.startswith("map_")}
    assert set(got) == set(want)
    for k in want:
        assert _same(got[k], want[k]), k
    assert want["errors"].tolist() == ["Duplicate frame warning", "Duplicate point3d warning", "No frame yet added",
                                       "No point yet added"]


# ---------------------------------------------------------------------------------------------------------------- BA graph + write-back
class _RecordingSolver:
    """Stands where Context.ba_solve stands: keeps the problem it is handed and returns it unchanged (the reference run used
    the identity in place of g2o's optimize())."""

    def __init__(self):
        self.problem = None

    def __call__(self, poses, pose_fixed, points, point_fixed, obs_pose, obs_point, obs_uv, K, huber_delta=0.0,
                 max_iterations=10, scale_edges=None, obs_info=None, dcs_phi=1.0):
        self.problem = dict(poses=np.array(poses), pose_fixed=np.array(pose_fixed), points=np.array(points),
                            point_fixed=np.array(point_fixed), obs_pose=np.array(obs_pose), obs_point=np.array(obs_point),
                            obs_uv=np.array(obs_uv), scale=scale_edges, huber=huber_delta, iters=max_iterations)
        return {"poses": np.array(poses), "points": np.array(points)}


@pytest.mark.parametrize("which", ["product", "ref_graph"])
@pytest.mark.parametrize("case", range(len(sc.GRAPH_CASES)))
def test_ba_problem_and_write_back_equal_the_executed_reference(fx, case, which):
    """The problem BundleAdjustment hands to the solver -- built from the map's structure-of-arrays mirror (product) or by
    oracle/ref_graph.py's restated double loop -- against the calls the reference's own localBundleAdjustement /
    motionOnlyBundleAdjustement made, and the map after the write-back."""
    from visual_slam_amd.LocalBA import BundleAdjustment, Camera
    from visual_slam_amd.map import Map
    from visual_slam_amd.point import Point
    from visual_slam_amd.workloads import ICL_NUIM_K
    name, kw = sc.GRAPH_CASES[case]
    pre = sc.graph_case_name(case) + "_"
    m, _, _ = sc.build_map(Map, Point, _product_frame, seed=31 + case)
    solver = _RecordingSolver()
    if which == "product":
        ba = BundleAdjustment(Camera(*ICL_NUIM_K), solver=solver)
    else:
        from oracle.ref_graph import RefLoopBundleAdjustment
        ba = RefLoopBundleAdjustment(Camera(*ICL_NUIM_K), solver=solver)
    getattr(ba, name)(m, **kw)
    p = solver.problem
    assert np.array_equal(p["poses"], fx[pre + "poses"]) and np.array_equal(p["pose_fixed"], fx[pre + "pose_fixed"])
    assert list(ba._pose_ids.keys()) == fx[pre + "pose_ids"].tolist()
    n_ref_points = len(fx[pre + "point_ids"])
    if "last_keyframe_id" in kw and which == "product":
        # the reference adds only the points visible to every frame; the product keeps every point in the arrays and fixes the
        # others without observations -- compare the free points and their edges
        free = np.nonzero(np.asarray(p["point_fixed"]) == 0)[0]
        ids = np.asarray(list(ba._point_ids.keys()))[free]
        assert ids.tolist() == fx[pre + "point_ids"].tolist()
        assert np.array_equal(p["points"][free], fx[pre + "points"])
        remap = -np.ones(len(p["points"]), np.int64)
        remap[free] = np.arange(len(free))
        assert np.array_equal(remap[p["obs_point"]], fx[pre + "obs_point"])
    else:
        assert list(ba._point_ids.keys()) == fx[pre + "point_ids"].tolist() and len(p["points"]) == n_ref_points
        assert np.array_equal(p["points"], fx[pre + "points"]) and np.array_equal(p["point_fixed"], fx[pre + "point_fixed"])
        assert np.array_equal(p["obs_point"], fx[pre + "obs_point"])
    assert np.array_equal(p["obs_pose"], fx[pre + "obs_pose"]) and np.array_equal(p["obs_uv"], fx[pre + "obs_uv"])
    assert p["huber"] == pytest.approx(np.sqrt(5.991)) and p["iters"] == 10
    if len(fx[pre + "scale_parent"]):
        sp, sch, sm = p["scale"]
        assert list(sp) == fx[pre + "scale_parent"].tolist() and list(sch) == fx[pre + "scale_child"].tolist()
        assert np.allclose(sm, fx[pre + "scale_meas"], rtol=0, atol=1e-15)
    else:
        assert not p["scale"]
    state = sc.map_state(m)
    assert np.allclose(state["poses_after"], fx[pre + "poses_after"], rtol=0, atol=1e-15)
    assert np.allclose(state["points_after"], fx[pre + "points_after"], rtol=0, atol=1e-15)


# ------------------------------------------------------------------------------------------------------------------- ratio loop + gathers
MF_CASES = range(4)


@pytest.mark.parametrize("case", MF_CASES)
def test_oracle_ratio_rule_equals_the_executed_reference_loop(fx, oracle, case):
    """The C oracle's k-NN + ratio test (the checker of ratio_compact_kernel) against the reference's own Lowe-ratio loop
    run on the NumPy twin's k-NN table."""
    pre = "mf%d_" % case
    idx, dist = oracle.hamming_knn2(fx[pre + "desc1"], fx[pre + "desc2"])
    assert np.array_equal(idx, fx[pre + "knn_idx"]) and np.array_equal(dist, fx[pre + "knn_dist"])
    mq, mt, md = oracle.match_ratio(fx[pre + "desc1"], fx[pre + "desc2"], float(fx[pre + "ratio"]))
    assert np.array_equal(mq, fx[pre + "query"]) and np.array_equal(mt, fx[pre + "train"])
    assert np.array_equal(md.astype(np.float64), fx[pre + "distance"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", MF_CASES)
def test_match_features_equals_the_executed_reference_loop(vs, fx, case):
    """FeatureMatcher.match_features on the GPU (hamming_knn2_kernel + ratio_compact_kernel + the host gathers) returns what
    the reference's own loop returned: the same [[DMatch]] rows and the same four gathered arrays, in the same order."""
    from visual_slam_amd.frame import FeatureMatcher
    pre = "mf%d_" % case
    fm = FeatureMatcher(context=vs)
    matches, pts1, ft1, pts2, ft2 = fm.match_features(fx[pre + "kp1"], fx[pre + "desc1"], fx[pre + "kp2"], fx[pre + "desc2"],
                                                      ratio=float(fx[pre + "ratio"]))
    assert all(len(m) == 1 for m in matches)
    assert [m[0].queryIdx for m in matches] == fx[pre + "query"].tolist()
    assert [m[0].trainIdx for m in matches] == fx[pre + "train"].tolist()
    assert [float(m[0].distance) for m in matches] == fx[pre + "distance"].tolist()
    for got, key in ((pts1, "pts1"), (ft1, "ft1"), (pts2, "pts2"), (ft2, "ft2")):
        want = fx[pre + key]
        assert np.array_equal(np.asarray(got).reshape(want.shape), want), key
