"""Two-view initialisation (helper_functions.py:47-70,164-195, main.py:88-148).

CPU: known answers of the oracle (exact E from noise-free samples, essential-manifold projection, RANSAC with outliers,
decomposition into the four candidates, cheirality vote, degenerate inputs) and the reference-shaped Python wrappers on
the oracle.  GPU: the HIP path against the oracle (same winner / masks, E and points to 1e-9).  OpenCV is not available:
parity with cv2.findEssentialMat / recoverPose is unpinned."""
import numpy as np
import pytest

from visual_slam_amd import helper_functions as hf
from visual_slam_amd.workloads import ICL_NUIM_K

K = np.array([[ICL_NUIM_K[0], 0, ICL_NUIM_K[2]], [0, ICL_NUIM_K[1], ICL_NUIM_K[3]], [0, 0, 1.0]])


def scene(n, outliers=0, noise_px=0.0, seed=0, rotvec=(0.02, -0.05, 0.01), t=(0.3, 0.02, -0.05)):
    from scipy.spatial.transform import Rotation
    r = np.random.default_rng(seed)
    X = np.stack([r.uniform(-2, 2, n), r.uniform(-1.5, 1.5, n), r.uniform(3, 8, n)], 1)
    R = Rotation.from_rotvec(rotvec).as_matrix()
    t = np.asarray(t, np.float64)
    x1 = X[:, :2] / X[:, 2:]
    Xc = (R @ X.T).T + t
    x2 = Xc[:, :2] / Xc[:, 2:] + (r.normal(0, noise_px / 480.0, (n, 2)) if noise_px else 0)
    out = np.sort(r.choice(n, outliers, replace=False))
    x2[out] += r.uniform(0.05, 0.2, (outliers, 2)) * r.choice([-1, 1], (outliers, 2))
    return x1, x2, R, t, X, out


def true_E(R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E) * np.sqrt(2)


def same_up_to_sign(A, B, tol):
    return min(np.abs(A - B).max(), np.abs(A + B).max()) < tol


def test_eight_point_is_exact_on_clean_data_and_lies_on_the_essential_manifold(oracle):
    x1, x2, R, t, _, _ = scene(40)
    for s in range(5):
        idx = np.random.default_rng(s).choice(40, 8, replace=False)
        E = oracle.eight_point(x1, x2, idx)
        assert same_up_to_sign(E, true_E(R, t), 1e-9)
        sv = np.linalg.svd(E)[1]
        assert abs(sv[0] - 1) < 1e-12 and abs(sv[1] - 1) < 1e-12 and sv[2] < 1e-12
    # a degenerate sample (one correspondence repeated) has a 2-dimensional null space -> rejected
    assert oracle.eight_point(x1, x2, np.array([0, 1, 2, 3, 4, 5, 6, 6], np.int32)) is None


def test_ransac_rejects_outliers_and_is_deterministic(oracle):
    x1, x2, R, t, _, out = scene(500, outliers=120, noise_px=0.5, seed=2)
    a = oracle.essential_ransac(x1, x2, 3.0 / 480, seed=4)
    b = oracle.essential_ransac(x1, x2, 3.0 / 480, seed=4)
    assert a["found"] and (a["mask"] == b["mask"]).all() and (a["E"] == b["E"]).all() and a["best_h"] == b["best_h"]
    assert a["n_inliers"] == a["mask"].sum() >= 0.9 * 380
    assert a["mask"][out].sum() <= 0.05 * 120            # the odd outlier may fall on its epipolar line
    assert 1 <= a["used"] < 1000                          # the budget shrank
    sv = np.linalg.svd(a["E"])[1]
    assert abs(sv[0] - 1) < 1e-9 and abs(sv[1] - 1) < 1e-9 and sv[2] < 1e-9
    clean = oracle.essential_ransac(*scene(300)[:2], 1e-6)
    assert clean["n_inliers"] == 300 and clean["used"] <= 2
    for n in (0, 5, 7):
        r = oracle.essential_ransac(x1[:n], x2[:n], 3.0 / 480)
        assert not r["found"] and r["n_inliers"] == 0 and not r["E"].any()


def test_decomposition_and_cheirality_vote(oracle):
    x1, x2, R, t, X, _ = scene(200, seed=5)
    E = true_E(R, t)
    R1, R2, tt = oracle.decompose_essential(E)
    for Q in (R1, R2):
        assert np.allclose(Q @ Q.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(Q) - 1) < 1e-12
    assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-9
    assert same_up_to_sign(tt, t / np.linalg.norm(t), 1e-9)
    for sign in (1.0, -1.0):                               # the vote does not depend on the sign of E
        rp = oracle.recover_pose(sign * E, x1, x2)
        assert rp["n_good"] == 200 and (rp["mask"] == 255).all()
        assert np.abs(rp["R"] - R).max() < 1e-9 and np.abs(rp["t"] - t / np.linalg.norm(t)).max() < 1e-9
        Xd = rp["X"][:, :3] / rp["X"][:, 3:]
        assert np.abs(Xd * np.linalg.norm(t) - X).max() < 1e-7     # structure up to the baseline scale
    far = oracle.recover_pose(E, x1, x2, dist_thresh=5.0 / np.linalg.norm(t))
    assert 0 < far["n_good"] < 200 and ((far["mask"] == 255) == (X[:, 2] < 5.0 - 1e-9) | (far["mask"] == 255) & (X[:, 2] < 5.0 + 1e-9)).all()
    empty = oracle.recover_pose(E, x1[:0], x2[:0])
    assert empty["n_good"] == 0 and empty["X"].shape == (0, 4)


def test_reference_shaped_wrappers_on_the_oracle(oracle):
    x1, x2, R, t, X, out = scene(400, outliers=80, noise_px=0.3, seed=5)
    to_px = lambda x: np.stack([x[:, 0] * K[0, 0] + K[0, 2], x[:, 1] * K[1, 1] + K[1, 2]], 1)
    p1, p2 = to_px(x1), to_px(x2)
    f1, f2 = np.arange(400)[:, None].repeat(32, 1).astype(np.uint8), np.arange(400)[:, None].repeat(32, 1).astype(np.uint8)
    E, inl, score = hf.estimateEssential(p1, p2, K, essTh=3.0 / K[0, 0], solver=oracle.essential_ransac)
    assert E.shape == (3, 3) and inl.shape == (400, 1) and inl.dtype == np.uint8 and set(np.unique(inl)) <= {0, 1}
    assert inl.sum() >= 300 and 7.9 * inl.sum() < score <= 8 * inl.sum()     # errors are ~1e-6 in normalised units
    sel = inl[:, 0] == 1
    Rr, tr, frac, Xh, q1, q2, g1, g2 = hf.estimateRelativePose(E, p1[sel], p2[sel], f1[sel], f2[sel], K, "Essential",
                                                               solver=oracle.recover_pose)
    assert frac > 0.95 and Xh.shape[0] == 4 and Xh.shape[1] == len(q1) == len(q2) == len(g1) == len(g2)
    assert tr.shape == (3, 1) and np.abs(Rr - R).max() < 0.02
    assert np.dot(tr[:, 0], t / np.linalg.norm(t)) > 0.95
    with pytest.raises(NotImplementedError):
        hf.estimateRelativePose(E, p1, p2, f1, f2, K, "Homography", solver=oracle.recover_pose)


@pytest.mark.gpu
@pytest.mark.parametrize("n,outliers,noise,seed", [(300, 0, 0.0, 0), (500, 120, 0.5, 2), (3000, 1200, 1.0, 3), (8, 0, 0.0, 4),
                                                   (9, 0, 0.2, 5)])
def test_gpu_essential_equals_oracle(vs, oracle, n, outliers, noise, seed):
    x1, x2, R, t, _, _ = scene(n, outliers, noise, seed)
    g = vs.essential_ransac(x1, x2, 3.0 / 480, seed=seed)
    c = oracle.essential_ransac(x1, x2, 3.0 / 480, seed=seed)
    assert g["found"] == c["found"] and (g["mask"] == c["mask"]).all()
    assert np.abs(g["E"] - c["E"]).max() < 1e-9


@pytest.mark.gpu
def test_gpu_recover_pose_equals_oracle(vs, oracle):
    for seed, noise in ((5, 0.0), (6, 0.5)):
        x1, x2, R, t, X, _ = scene(700, 0, noise, seed)
        E = oracle.essential_ransac(x1, x2, 3.0 / 480, seed=seed)["E"] if noise else true_E(R, t)
        g, c = vs.recover_pose(E, x1, x2), oracle.recover_pose(E, x1, x2)
        assert g["n_good"] == c["n_good"] and (g["mask"] == c["mask"]).all()
        assert np.abs(g["R"] - c["R"]).max() < 1e-12 and np.abs(g["t"] - c["t"]).max() < 1e-12
        assert np.abs(g["X"] - c["X"]).max() < 1e-9
    e = vs.recover_pose(true_E(R, t), x1[:0], x2[:0])
    assert e["n_good"] == 0 and e["X"].shape == (0, 4)
    r = vs.essential_ransac(x1[:5], x2[:5], 1e-3)
    assert not r["found"] and r["n_inliers"] == 0
