"""NumPy restatements of the parts of the reference that are pure NumPy themselves.  TEST INFRASTRUCTURE ONLY.

Unlike oracle/vs_oracle.c (whose counterparts live inside cv2 / g2o and cannot be pinned), these functions restate
arithmetic the reference performs with NumPy itself, so they run HERE with the reference's own numerical back end
(np.linalg.svd = LAPACK gesdd): parity for these rows is PINNED to the reference's algorithm.  The reference module
cannot be imported (it imports cv2 at the top, ModuleNotFoundError), hence the restatement.
"""
import numpy as np


def triangulate(pose1, pose2, pts1, pts2):
    """reference src/v2/helper_functions.py:281-291 -- one 4x4 DLT system per match, X = last right singular vector."""
    ret = np.zeros((pts1.shape[0], 4))
    for i, p in enumerate(zip(pts1, pts2)):
        A = np.zeros((4, 4))
        A[0] = p[0][0] * pose1[2] - pose1[0]
        A[1] = p[0][1] * pose1[2] - pose1[1]
        A[2] = p[1][0] * pose2[2] - pose2[0]
        A[3] = p[1][1] * pose2[2] - pose2[1]
        _, _, vt = np.linalg.svd(A)
        ret[i] = vt[3]
    return ret


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
