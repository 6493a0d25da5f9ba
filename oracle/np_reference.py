"""NumPy restatements of the parts of the reference that are pure NumPy themselves.  TEST INFRASTRUCTURE ONLY.

Unlike oracle/vs_oracle.c (whose counterparts live inside cv2 / g2o and cannot be pinned), these functions restate
arithmetic the reference performs with NumPy itself, so they run HERE with the reference's own numerical back end
(np.linalg.svd = LAPACK gesdd): parity for these rows is PINNED to the reference's algorithm.  The reference module
cannot be imported (it imports cv2 at the top, ModuleNotFoundError), hence the restatement.
"""
import numpy as np


def triangulate(pose1, pose2, pts1, pts2):
    """What reference src/v2/helper_functions.py:281-291 computes -- per match the right singular vector of the smallest singular
    value of the 4x4 DLT system [u1*P1[2]-P1[0]; v1*P1[2]-P1[1]; u2*P2[2]-P2[0]; v2*P2[2]-P2[1]] -- as ONE batched
    decomposition.  Checked against the stored outputs of the reference's own function (tests/test_ref_fixtures.py); the GPU
    tests compare the kernel with those stored outputs, this function serves the CPU driver (tests/test_slam_driver.py)."""
    pts1, pts2 = np.asarray(pts1, np.float64), np.asarray(pts2, np.float64)
    if len(pts1) == 0:
        return np.zeros((0, 4))
    uv = np.concatenate([pts1[:, :2], pts2[:, :2]], axis=1)                    # [n, 4] = u1 v1 u2 v2
    last = np.stack([pose1[2], pose1[2], pose2[2], pose2[2]])                   # row multiplied by the image coordinate
    first = np.stack([pose1[0], pose1[1], pose2[0], pose2[1]])                  # row subtracted
    systems = uv[:, :, None] * last[None] - first[None]                         # [n, 4, 4]
    return np.linalg.svd(systems)[2][:, 3, :]


def cheirality_filter(p1, p2, X4):
    """reference src/v2/main.py:286-309: dehomogenise, depth in both cameras, keep 0 < z < 1 in both."""
    X = X4 / X4[:, 3:]
    proj1 = p1 @ X.T
    proj2 = p2 @ X.T
    good = np.where((proj1[2] > 0) & (proj2[2] > 0) & (proj2[2] < 1) & (proj1[2] < 1))[0]
    return X[:, :3], good, np.stack([proj1[2], proj2[2]], 1)


def make_homogeneous(x):
    """reference src/v2/helper_functions.py:362-364"""
    return np.concatenate((x, np.ones((len(x), 1))), axis=1)


def camera_projection_matrix2(pose, K):
    """reference src/v2/helper_functions.py:376-377"""
    return K @ pose[0:3, :]
