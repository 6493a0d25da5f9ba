"""Dataset I/O and trajectory evaluation.

The reference ships the files but never reads them: its main.py globs the rgb directory (main.py:66-76) and reports no
error metric.  Formats (data/ICL_NUIM):
  associations.txt      `idx depth/N.png idx rgb/N.png` per line
  traj3.gt.freiburg     `idx tx ty tz qx qy qz qw` per line (TUM trajectory format, camera-to-world, idx from 1)
Host-side NumPy only - this runs once per sequence, nothing here is on the per-frame path.
"""
import os

import numpy as np

from .frame import imread
from .workloads import ICL_NUIM_K


def read_associations(path):
    """-> list of (depth_index, depth_relpath, rgb_index, rgb_relpath); malformed or comment lines are skipped."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) != 4 or line.lstrip().startswith("#"):
                continue
            out.append((int(p[0]), p[1], int(p[2]), p[3]))
    return out


def quat_to_matrix(q):
    """(qx, qy, qz, qw) -> 3x3 rotation (normalised first)."""
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def read_trajectory(path):
    """TUM trajectory file -> (indices int[n], poses float64[n,4,4] camera-to-world)."""
    idx, poses = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) != 8 or line.lstrip().startswith("#"):
                continue
            v = [float(t) for t in p[1:]]
            T = np.eye(4)
            T[:3, :3] = quat_to_matrix(v[3:7])
            T[:3, 3] = v[0:3]
            idx.append(int(float(p[0])))
            poses.append(T)
    return np.asarray(idx, np.int64), np.asarray(poses).reshape(-1, 4, 4)


def write_trajectory(path, indices, poses):
    """poses [n,4,4] camera-to-world -> TUM format (the inverse of read_trajectory)."""
    from scipy.spatial.transform import Rotation
    with open(path, "w") as f:
        for i, T in zip(indices, poses):
            q = Rotation.from_matrix(np.asarray(T)[:3, :3]).as_quat()
            t = np.asarray(T)[:3, 3]
            f.write("%d %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n" % (i, t[0], t[1], t[2], q[0], q[1], q[2], q[3]))


class Sequence:
    """A TUM / ICL-NUIM style directory: rgb/, depth/, optional associations.txt and ground truth."""

    def __init__(self, root, associations="associations.txt", groundtruth=None, K=ICL_NUIM_K, depth_scale=5000.0):
        self.root, self.K, self.depth_scale = root, K, depth_scale
        a = os.path.join(root, associations)
        if os.path.exists(a):
            self.frames = read_associations(a)
        else:  # main.py:66-76: natural order of the rgb directory
            names = sorted(os.listdir(os.path.join(root, "rgb")), key=lambda s: int(os.path.splitext(s)[0]))
            self.frames = [(int(os.path.splitext(n)[0]), "depth/" + n, int(os.path.splitext(n)[0]), "rgb/" + n) for n in names]
        self.gt = read_trajectory(os.path.join(root, groundtruth)) if groundtruth else None

    def __len__(self):
        return len(self.frames)

    def rgb(self, i):
        return imread(os.path.join(self.root, self.frames[i][3]))

    def depth(self, i):
        from PIL import Image
        return np.asarray(Image.open(os.path.join(self.root, self.frames[i][1]))).astype(np.float64) / self.depth_scale


def umeyama(src, dst, with_scale=True):
    """Least-squares similarity dst ~ s R src + t (Umeyama 1991).  src, dst [n,3].  -> (s, R, t)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    var = (xs ** 2).sum() / len(src)
    s = float(np.trace(np.diag(D) @ S) / var) if with_scale and var > 0 else 1.0
    return s, R, mu_d - s * R @ mu_s


def ate_rmse(est_poses, gt_poses, with_scale=True):
    """Absolute trajectory error after Sim(3) (monocular: the scale is free) or SE(3) alignment of the camera centres.
    -> dict(rmse, mean, max, scale, path_length)."""
    e = np.asarray(est_poses)[:, :3, 3]
    g = np.asarray(gt_poses)[:, :3, 3]
    s, R, t = umeyama(e, g, with_scale)
    d = np.linalg.norm((s * (R @ e.T)).T + t - g, axis=1)
    return dict(rmse=float(np.sqrt((d ** 2).mean())), mean=float(d.mean()), max=float(d.max()), scale=s,
                path_length=float(np.linalg.norm(np.diff(g, axis=0), axis=1).sum()))
