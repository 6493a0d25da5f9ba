"""The subset of the reference's src/v2/helper_functions.py that sits next to the hot path, without cv2.

triangulate() runs on the GPU (vs_triangulate_dlt); the rest are the reference's small NumPy helpers, kept so that the
key-frame code of main.py:237-318 finds the names it imports.  RANSAC geometry (estimateEssential, estimateHomography,
estimateRelativePose: cv2.findEssentialMat / recoverPose) is out of scope this round.
"""
import math as _math

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
    if kp1.dtype == np.float32 and kp2.dtype == np.float32 and kp1.shape[1] >= 2 and kp2.shape[1] >= 2:
        # key points as FeatureExtractor hands them out: a row is two float32 = ONE 64-bit word, and equal words are equal rows
        # (+0.0 folds -0.0 onto +0.0; a NaN never equals anything in the reference's loop either: such rows stay "not found")
        a = np.ascontiguousarray(kp1[:, :2] + np.float32(0.0)).view(np.uint64).ravel()
        b = np.ascontiguousarray(kp2[:, :2] + np.float32(0.0)).view(np.uint64).ravel()
        bs = np.sort(b)                                  # (np.isin costs three times this on a few hundred rows)
        pos = np.searchsorted(bs, a)
        pos[pos == bs.size] = bs.size - 1
        found = bs[pos] == a
        if np.isnan(kp1[:, :2]).any():
            found &= ~np.isnan(kp1[:, :2]).any(axis=1)
        return np.nonzero(~found)[0].tolist()
    a = kp1[:, :2].astype(np.float64).view(np.complex128).ravel()
    b = kp2[:, :2].astype(np.float64).view(np.complex128).ravel()
    return np.nonzero(~np.isin(a, b))[0].tolist()


def _quat_to_rvec(w, x, y, z):
    if w < 0.0:
        w, x, y, z = -w, -x, -y, -z
    n = _math.sqrt(x * x + y * y + z * z)
    k = 2.0 if n < 1e-12 else 2.0 * _math.atan2(n, w) / n  # angle / sin(angle / 2) -> 2 for small angles
    return k * x, k * y, k * z


def _rvec_of_rows(m):
    """rotation matrix as nested lists of Python floats -> rotation vector (x, y, z).  Through the unit quaternion
    (largest-component branch, as Eigen / scipy do): accurate near 0 and near pi.  Plain float arithmetic: the tracking
    loop calls this twice per frame, and NumPy scalars cost a microsecond per operation."""
    (m00, m01, m02), (m10, m11, m12), (m20, m21, m22) = m
    tr = m00 + m11 + m22
    if tr > 0.0:
        s = _math.sqrt(tr + 1.0) * 2.0
        return _quat_to_rvec(0.25 * s, (m21 - m12) / s, (m02 - m20) / s, (m10 - m01) / s)
    if m00 > m11 and m00 > m22:
        s = _math.sqrt(1.0 + m00 - m11 - m22) * 2.0
        return _quat_to_rvec((m21 - m12) / s, 0.25 * s, (m01 + m10) / s, (m02 + m20) / s)
    if m11 > m22:
        s = _math.sqrt(1.0 + m11 - m00 - m22) * 2.0
        return _quat_to_rvec((m02 - m20) / s, (m01 + m10) / s, 0.25 * s, (m12 + m21) / s)
    s = _math.sqrt(1.0 + m22 - m00 - m11) * 2.0
    return _quat_to_rvec((m10 - m01) / s, (m02 + m20) / s, (m12 + m21) / s, 0.25 * s)


def Rtorvec(R):
    """helper_functions.py:276-278 (cv2.Rodrigues(R)[0]): rotation matrix -> rotation vector [3,1]."""
    x, y, z = _rvec_of_rows(np.asarray(R, np.float64).reshape(3, 3).tolist())
    return np.array([[x], [y], [z]])


def _rodrigues_rows(x, y, z):
    """rotation vector -> rotation matrix rows (Python floats): I + a K + b K^2, K^2 = r r^T - |r|^2 I."""
    th2 = x * x + y * y + z * z
    th = _math.sqrt(th2)
    if th < 1e-12:
        a, b = 1.0, 0.5  # sin(th)/th, (1 - cos(th))/th^2
    else:
        a, b = _math.sin(th) / th, (1.0 - _math.cos(th)) / th2
    ax, ay, az = a * x, a * y, a * z
    bxy, bxz, byz = b * x * y, b * x * z, b * y * z
    return ([1.0 - b * (y * y + z * z), bxy - az, bxz + ay],
            [bxy + az, 1.0 - b * (x * x + z * z), byz - ax],
            [bxz - ay, byz + ax, 1.0 - b * (x * x + y * y)])


def transformMatrix(rvec, tvec):
    """helper_functions.py:269-274: 4x4 from a rotation vector (Rodrigues) and a translation."""
    x, y, z = np.asarray(rvec, np.float64).reshape(3).tolist()
    tx, ty, tz = np.asarray(tvec, np.float64).reshape(3).tolist()
    r0, r1, r2 = _rodrigues_rows(x, y, z)
    # the reference returns np.matrix (main.py:202 slices it and squeezes with np.asarray)
    return np.array([r0 + [tx], r1 + [ty], r2 + [tz], [0.0, 0.0, 0.0, 1.0]]).view(np.matrix)


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
        x, y, z = np.asarray(rvec, np.float64).reshape(3).tolist()
        tx, ty, tz = np.asarray(tvec, np.float64).reshape(3).tolist()
        (a, b, c), (d, e, f), (g, h, i) = _rodrigues_rows(x, y, z)  # world-to-camera rotation; the pose is its inverse
        pose0 = np.array([[a, d, g, -(a * tx + d * ty + g * tz)], [b, e, h, -(b * tx + e * ty + h * tz)],
                          [c, f, i, -(c * tx + f * ty + i * tz)], [0.0, 0.0, 0.0, 1.0]])
    else:
        pose0 = np.eye(4)
    r = None
    if solver is None:
        ctx = context or default_context()
        own = ctx._track_owner
        if own is not None and own.spec is not None and useExtrinsicGuess:
            # the correspondences are the matches the resident period just produced and the guess is its previous pose:
            # PnP-RANSAC runs there (no upload), with the motion-only BA enqueued right behind it
            r = own.speculate_back(obj, img, K4, pose0, int(iterationsCount), float(reprojectionError), float(confidence),
                                   int(seed))
    if r is None:
        run = solver or (context or default_context()).pnp_ransac
        r = run(obj, img, K4, pose0, iterations=int(iterationsCount), reproj_err=float(reprojectionError),
                confidence=float(confidence), seed=int(seed))
    (a, b, c, px), (d, e, f, py), (g, h, i, pz) = np.asarray(r["pose"], np.float64)[:3].tolist()  # camera-to-world
    x, y, z = _rvec_of_rows([[a, d, g], [b, e, h], [c, f, i]])                                       # ... inverted
    tvec_out = np.array([[-(a * px + d * py + g * pz)], [-(b * px + e * py + h * pz)], [-(c * px + f * py + i * pz)]])
    return bool(r["found"]), np.array([[x], [y], [z]]), tvec_out, r["inliers"].astype(np.int32).reshape(-1, 1)


def _normalise(pts, K):
    """cv2.undistortPoints(pts, K, None) without distortion: pixel -> K-normalised coordinates."""
    K = np.asarray(K, np.float64)
    pts = np.asarray(pts, np.float64).reshape(-1, 2)
    return np.stack([(pts[:, 0] - K[0, 2]) / K[0, 0], (pts[:, 1] - K[1, 2]) / K[1, 1]], 1)


def matlab_max(v, s):
    """helper_functions.py:43-44."""
    return [max(v[i], s) for i in range(len(v))]


def estimateEssential(pts1, pts2, K, essTh, context=None, seed=0, solver=None):
    """helper_functions.py:47-70 -> (E, inliers uint8 [N,1] (1/0), score).

    cv2.findEssentialMat(RANSAC, prob=0.999, threshold=essTh) on the K-normalised points is replaced by
    Context.essential_ransac (8-point hypotheses); the score is the reference's own NumPy: squared distances
    to the epipolar lines (cv2.computeCorrespondEpilines == l = E^T x2 resp. E x1, scaled to a^2 + b^2 = 1) of the
    inliers in both images against outlierThreshold = 4.  `solver(x1, x2, threshold, seed=)` injects the CPU oracle."""
    x1, x2 = _normalise(pts1, K), _normalise(pts2, K)
    run = solver or (context or default_context()).essential_ransac
    r = run(x1, x2, float(essTh), seed=int(seed))
    E = r["E"]
    inliers = r["mask"].astype(np.uint8).reshape(-1, 1)
    sel = inliers[:, 0] == 1
    p1, p2 = x1[sel], x2[sel]
    loc1 = np.concatenate((p1, np.ones((len(p1), 1))), axis=1)
    loc2 = np.concatenate((p2, np.ones((len(p2), 1))), axis=1)

    def epilines(E_, pts_h):  # computeCorrespondEpilines: l = E_ x, normalised so that a^2 + b^2 = 1
        l = pts_h @ E_.T
        nrm = np.sqrt(l[:, 0] ** 2 + l[:, 1] ** 2)
        nrm[nrm == 0] = 1.0
        return l / nrm[:, None]

    line_in_1 = epilines(E.T, loc2)  # whichImage=2 -> lines in image 1
    err_2in1 = np.sum(loc1 * line_in_1, axis=1) ** 2 / np.sum(line_in_1[:, :3] ** 2, axis=1)
    line_in_2 = epilines(E.T, loc1)  # the reference passes whichImage=2 for this direction as well (:66)
    err_1in2 = np.sum(loc2 * line_in_2, axis=1) ** 2 / np.sum(line_in_2[:, :3] ** 2, axis=1)
    outlier_threshold = 4
    score = np.sum(matlab_max(outlier_threshold - err_1in2, 0)) + sum(matlab_max(outlier_threshold - err_2in1, 0))
    return E, inliers, score


def estimateRelativePose(tform, inlier_pts1, inlier_pts2, inlier_fts1, inlier_fts2, K, tform_type="Essential",
                         context=None, solver=None):
    """helper_functions.py:164-191 (Essential branch): cv2.recoverPose(E, pts1, pts2, K, distanceThresh=50) and the
    gather of the points it accepted -> (R, t [3,1], validFraction, X [4,M], pts1, pts2, fts1, fts2)."""
    if tform_type != "Essential":
        raise NotImplementedError("estimateRelativePose: only the Essential branch is live in the reference (main.py:109)")
    x1, x2 = _normalise(inlier_pts1, K), _normalise(inlier_pts2, K)
    run = solver or (context or default_context()).recover_pose
    r = run(np.asarray(tform, np.float64), x1, x2, 50.0)
    good = np.where(r["mask"] == 255)[0]
    n = max(len(x1), 1)
    X = r["X"][good].T if len(good) else np.zeros((4, 0))
    return (r["R"], r["t"].reshape(3, 1), len(good) / n, X, [inlier_pts1[i] for i in good],
            [inlier_pts2[i] for i in good], [inlier_fts1[i] for i in good], [inlier_fts2[i] for i in good])
