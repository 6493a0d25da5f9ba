"""The subset of the reference's src/v2/helper_functions.py that sits next to the hot path, without cv2.

triangulate() runs on the GPU (vs_triangulate_dlt); the rest are the reference's small NumPy helpers, kept so that the
key-frame code of main.py:237-318 finds the names it imports.  RANSAC geometry (estimateEssential, estimateHomography,
estimateRelativePose: cv2.findEssentialMat / recoverPose) is out of scope this round.
"""
import numpy as np

from .context import default_context


def MakeHomogeneous(x):
    """helper_functions.py:362-364"""
    col_of_ones = np.ones((len(x), 1))
    return np.concatenate((x, col_of_ones), axis=1)


def CameraProjectionMatrix2(Pose, K):
    """helper_functions.py:376-377: K @ Pose[0:3, :] (Pose = world-to-camera 4x4)."""
    return np.asarray(K) @ np.asarray(Pose)[0:3, :]


def CameraProjectionMatrix(R, t, K):
    """helper_functions.py:367-371"""
    rotation_mat = R.T
    translationVec = -t.T @ R.T
    return (np.concatenate((rotation_mat, translationVec), axis=0) @ K).T


def triangulate(pose1, pose2, pts1, pts2, context=None):
    """helper_functions.py:281-291 on the GPU.  pose1/pose2: 3x4 projection matrices; pts1/pts2: [N, >=2] (the reference
    passes homogeneous pixels).  Returns [N, 4] homogeneous points, unit norm, w >= 0 (the reference returns LAPACK's
    arbitrary sign; its caller divides by w in the next line, main.py:286)."""
    ctx = context or default_context()
    return np.array(ctx.triangulate_dlt(pose1, pose2, pts1, pts2))


def triangulate_and_filter(p1, p2, K, pts1, pts2, context=None):
    """main.py:263-309 in one call: projection matrices K @ p[0:3], triangulation, dehomogenisation and the cheirality /
    depth filter `(z1 > 0) & (z2 > 0) & (z2 < 1) & (z1 < 1)`.  p1, p2: world-to-camera 4x4.
    Returns (points [N,3], indices of the good points)."""
    ctx = context or default_context()
    X4, depth = ctx.triangulate_dlt(CameraProjectionMatrix2(p1, K), CameraProjectionMatrix2(p2, K), pts1, pts2, p1, p2)
    X = X4[:, :3] / X4[:, 3:]
    good = np.where((depth[:, 0] > 0) & (depth[:, 1] > 0) & (depth[:, 1] < 1) & (depth[:, 0] < 1))[0]
    return X, good


def GetListDiff(kp1, kp2):
    """helper_functions.py:316-326: indices of the rows of kp1 that do not occur in kp2.  (The reference computes this
    with an O(N*M) Python loop, discards the result and re-computes it with `x not in kp2`; for 2-D arrays that
    membership test is row-wise "any element equal", which the loop version -- both coordinates equal -- corrects.
    The loop semantics are implemented, vectorised.)"""
    kp1 = np.asarray(kp1)
    kp2 = np.asarray(kp2)
    if kp1.size == 0:
        return []
    if kp2.size == 0:
        return list(range(len(kp1)))
    a = kp1[:, :2].astype(np.float64).view(np.complex128).ravel()
    b = kp2[:, :2].astype(np.float64).view(np.complex128).ravel()
    return np.nonzero(~np.isin(a, b))[0].tolist()


def Rtorvec(R):
    """helper_functions.py:276-278 (cv2.Rodrigues(R)[0]): rotation matrix -> rotation vector [3,1]."""
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R, np.float64)).as_rotvec().reshape(3, 1)


def transformMatrix(rvec, tvec):
    """helper_functions.py:269-274: 4x4 from a rotation vector (Rodrigues) and a translation."""
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(np.asarray(rvec, np.float64).reshape(3)).as_matrix()
    T[:3, 3] = np.asarray(tvec, np.float64).reshape(3)
    return np.matrix(T)  # the reference returns np.matrix (main.py:202 slices it and squeezes with np.asarray)


def solvePnPRansac(objectPoints, imagePoints, cameraMatrix, distCoeffs=None, rvec=None, tvec=None,
                   useExtrinsicGuess=False, iterationsCount=100, reprojectionError=8.0, confidence=0.99, inliers=None,
                   flags=None, context=None, seed=0, solver=None):
    """cv2.solvePnPRansac as main.py:196-197 calls it -> (retval, rvec [3,1], tvec [3,1], inliers [M,1] int32).

    rvec/tvec are OpenCV's world-to-camera transform.  Hypotheses are LM refinements of the extrinsic guess on 5 sampled
    correspondences (OpenCV's ITERATIVE solver with useExtrinsicGuess); without a guess the start is the identity
    (OpenCV would run a DLT first - not provided).  distCoeffs must be empty.  `solver(obj, img, K4, pose0, ...)`
    replaces Context.pnp_ransac (tests inject the CPU oracle)."""
    if distCoeffs is not None and np.asarray(distCoeffs).size and np.any(np.asarray(distCoeffs) != 0):
        raise ValueError("solvePnPRansac: lens distortion is not supported")
    obj = np.asarray(objectPoints, np.float64).reshape(-1, 3)
    img = np.asarray(imagePoints, np.float64).reshape(-1, 2)
    Kmat = np.asarray(cameraMatrix, np.float64)
    K4 = (Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2])
    if useExtrinsicGuess and rvec is not None and tvec is not None:
        c_T_w = np.asarray(transformMatrix(rvec, np.asarray(tvec, np.float64).reshape(3, 1)))
    else:
        c_T_w = np.eye(4)
    pose0 = np.eye(4)
    pose0[:3, :3] = c_T_w[:3, :3].T
    pose0[:3, 3] = -c_T_w[:3, :3].T @ c_T_w[:3, 3]
    run = solver or (context or default_context()).pnp_ransac
    r = run(obj, img, K4, pose0, iterations=int(iterationsCount), reproj_err=float(reprojectionError),
            confidence=float(confidence), seed=int(seed))
    w_T_c = r["pose"]
    R = w_T_c[:3, :3].T
    t = -R @ w_T_c[:3, 3]
    return bool(r["found"]), Rtorvec(R), t.reshape(3, 1), r["inliers"].astype(np.int32).reshape(-1, 1)
