"""Two-view DLT triangulation.  The reference implements it in pure NumPy
(src/v2/helper_functions.py:281-291): the GPU tests compare vs_triangulate_dlt with what the reference's OWN function returned
for these scenes (tests/golden/ref_fixtures.npz, tri_tv<i>_*, written by tests/golden/make_ref_fixtures.py) -- parity pinned.
Tolerance: 1e-9 relative on the dehomogenised points (two different SVD algorithms in FP64; the DLT systems of a
sane two-view geometry have singular-value gaps of 1e-3..1e-1)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import np_reference as ref
from ref_scenarios import TWO_VIEW_CASES
from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "ref_fixtures.npz"))


def _stored(fx, n, noise, seed):
    """The stored scene (n, noise, seed) -- its inputs are regenerated here and must equal the stored ones bit for bit -- and
    the reference's output for it."""
    pre = "tri_tv%d_" % TWO_VIEW_CASES.index((n, noise, seed))
    K, p1, p2, x1, x2, gt = _two_view(n, seed, noise)
    assert np.array_equal(x1, fx[pre + "x1"]) and np.array_equal(x2, fx[pre + "x2"]) and np.array_equal(p2, fx[pre + "w2c2"])
    return K, p1, p2, x1, x2, gt, fx[pre + "P1"], fx[pre + "P2"], fx[pre + "X"]


def _two_view(n=500, seed=0, noise=0.0):
    w = ba_workload(n_cams=2, n_points=n, seed=seed, noise_px=noise, outlier_frac=0, pose_sigma_t=0, pose_sigma_deg=0,
                    point_sigma=0)
    fx, fy, cx, cy = ICL_NUIM_K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    p1, p2 = np.linalg.inv(w["poses_gt"][0]), np.linalg.inv(w["poses_gt"][1])  # world -> camera (main.py:263-264)
    uv = w["obs_uv"].reshape(n, 2, 2)
    return K, p1, p2, ref.make_homogeneous(uv[:, 0]), ref.make_homogeneous(uv[:, 1]), w["points_gt"]


def test_reference_restatement_recovers_noise_free_points():
    K, p1, p2, x1, x2, gt = _two_view()
    X4 = ref.triangulate(ref.camera_projection_matrix2(p1, K), ref.camera_projection_matrix2(p2, K), x1, x2)
    X, good, depth = ref.cheirality_filter(p1, p2, X4)
    assert np.allclose(X, gt, atol=1e-8)
    assert np.allclose(np.linalg.norm(X4, axis=1), 1.0)
    assert len(good) == 0 and depth.min() > 2.0  # main.py keeps 0 < z < 1 only: nothing in this 2.5-5.5 m scene


def test_host_helpers_match_the_reference_definitions():
    from visual_slam_amd import helper_functions as hf
    K, p1, p2, x1, x2, _ = _two_view(10)
    assert np.array_equal(hf.MakeHomogeneous(x1[:, :2]), ref.make_homogeneous(x1[:, :2]))
    assert np.array_equal(hf.CameraProjectionMatrix2(p1, K), ref.camera_projection_matrix2(p1, K))
    kp1 = np.array([[1.0, 2], [3, 4], [5, 6], [1, 6]], np.float32)
    kp2 = np.array([[3.0, 4], [9, 9], [1, 2]], np.float32)
    assert hf.GetListDiff(kp1, kp2) == [2, 3]
    assert hf.GetListDiff(kp1, kp2[:0]) == [0, 1, 2, 3] and hf.GetListDiff(kp1[:0], kp2) == []
    R = p2[:3, :3]
    T = hf.transformMatrix(hf.Rtorvec(R), p2[:3, 3])
    assert np.allclose(np.asarray(T), p2, atol=1e-12) and hf.Rtorvec(R).shape == (3, 1)


def test_stored_scenes_are_the_scenes_generated_here(fx):
    for n, noise, seed in TWO_VIEW_CASES:
        K, p1, p2, x1, x2, gt, P1, P2, want = _stored(fx, n, noise, seed)
        assert np.array_equal(ref.camera_projection_matrix2(p1, K), P1) and np.array_equal(ref.camera_projection_matrix2(p2, K), P2)
        assert want.shape == (n, 4) and np.allclose(np.linalg.norm(want, axis=1), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,noise,seed", TWO_VIEW_CASES[:4])
def test_hip_triangulation_equals_the_reference_output(vs, fx, n, noise, seed):
    K, p1, p2, x1, x2, gt, P1, P2, want = _stored(fx, n, noise, seed)
    X4, depth = vs.triangulate_dlt(P1, P2, x1, x2, p1, p2)
    assert np.allclose(np.linalg.norm(X4, axis=1), 1.0, atol=1e-14) and np.all(X4[:, 3] >= 0)
    want_s = want * np.where(want[:, 3:] < 0, -1.0, 1.0)  # LAPACK's sign is arbitrary
    assert np.allclose(X4, want_s, rtol=0, atol=1e-11)
    Xg, Xw = X4[:, :3] / X4[:, 3:], want[:, :3] / want[:, 3:]
    assert np.max(np.linalg.norm(Xg - Xw, axis=1) / np.linalg.norm(Xw, axis=1)) < 1e-9
    _, _, wdepth = ref.cheirality_filter(p1, p2, want)  # main.py:286-309 applied to the reference's points
    assert np.allclose(depth, wdepth, rtol=1e-9, atol=1e-9)
    if noise == 0.0:
        assert np.allclose(Xg, gt, atol=1e-8)


@pytest.mark.gpu
def test_helper_functions_triangulate_drop_in(vs, fx):
    from visual_slam_amd import helper_functions as hf
    K, p1, p2, x1, x2, gt, _, _, want = _stored(fx, 800, 0.3, 7)
    P1, P2 = hf.CameraProjectionMatrix2(p1, K), hf.CameraProjectionMatrix2(p2, K)
    pts = hf.triangulate(pose1=P1, pose2=P2, pts1=x1, pts2=x2, context=vs)   # main.py:284
    pts /= pts[:, 3:]                                                       # main.py:286
    want = want / want[:, 3:]
    assert np.allclose(pts, want, rtol=1e-9, atol=1e-9)
    # the fused call applies main.py's filter; scale the scene so that some depths fall inside (0, 1).  (The scaled scene is not
    # among the stored ones: its checker is oracle/np_reference.py, itself held to the stored outputs by test_ref_fixtures.py.)
    s = 0.2
    p1s, p2s = p1.copy(), p2.copy()
    p1s[:3, 3] *= s
    p2s[:3, 3] *= s
    X, good = hf.triangulate_and_filter(p1s, p2s, K, x1, x2, context=vs)
    Xr, goodr, _ = ref.cheirality_filter(p1s, p2s, ref.triangulate(hf.CameraProjectionMatrix2(p1s, K), hf.CameraProjectionMatrix2(p2s, K), x1, x2))
    assert np.array_equal(good, goodr) and 0 < len(good) < len(X) and np.allclose(X, Xr, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_degenerate_inputs(vs):
    K, p1, p2, x1, x2, _ = _two_view(5)
    P1 = ref.camera_projection_matrix2(p1, K)
    assert vs.triangulate_dlt(P1, P1, x1[:0], x2[:0]).shape == (0, 4)
    X4 = vs.triangulate_dlt(P1, P1, x1, x1)  # identical views: rank-deficient system, still a finite unit vector
    assert np.all(np.isfinite(X4)) and np.allclose(np.linalg.norm(X4, axis=1), 1.0)
