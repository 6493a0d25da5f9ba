"""Seeded synthetic workloads of BASELINE.json's configs.  Shared by bench.py and tests/.

Nothing here touches /root/reference or the oracle; everything is regenerated from fixed seeds.
"""
import numpy as np

ICL_NUIM_K = (481.20, 480.0, 319.5, 239.5)  # fx, fy, cx, cy (reference src/v1/slam_test.py:144-145, main.py:55-56)


def match_workload(n_query=10000, n_train=10000, n_dup=64, seed=0):
    """cfg3 / cfg5: train = random 256-bit rows; 70 % of the queries are a permuted train row with each bit flipped
    w.p. 0.05, 30 % are fresh random rows; n_dup exact duplicate rows are planted in train to exercise the tie rule
    (lower train index wins)."""
    rng_t = np.random.default_rng(seed)
    train = rng_t.integers(0, 256, (n_train, 32), dtype=np.uint8)
    if n_dup and n_train >= 2 * n_dup:
        src = rng_t.choice(n_train, size=n_dup, replace=False)
        dst = (src + n_train // 2) % n_train
        train[dst] = train[src]
    rng_q = np.random.default_rng(seed + 1)
    perm = rng_q.permutation(n_train)
    pick = perm[np.arange(n_query) % n_train]
    query = train[pick].copy()
    flips = rng_q.random((n_query, 256)) < 0.05
    query ^= np.packbits(flips, axis=1, bitorder="little")
    fresh = rng_q.random(n_query) < 0.30
    query[fresh] = rng_q.integers(0, 256, (int(fresh.sum()), 32), dtype=np.uint8)
    return np.ascontiguousarray(query), np.ascontiguousarray(train)


def synthetic_frame(w=640, h=480, seed=2):
    """cfg2's synthetic frame: uniform noise box-blurred 5x5 so FAST sees a realistic corner density."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h + 4, w + 4), dtype=np.uint8).astype(np.uint32)
    acc = np.zeros((h, w), np.uint32)
    for dy in range(5):
        for dx in range(5):
            acc += img[dy:dy + h, dx:dx + w]
    g = (acc // 25).astype(np.int32)
    # stretch contrast so the blurred noise still has corners above threshold 20
    g = np.clip((g - 128) * 6 + 128, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))


def _rot(axis, ang):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)


def _look_at(cam_pos, target):
    """camera-to-world rotation with +z towards target, +y down-ish (image convention)."""
    z = target - cam_pos
    z = z / np.linalg.norm(z)
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(-up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)


def ba_workload(n_cams=10, n_points=2000, seed=3, noise_px=0.5, outlier_frac=0.02, visibility=1.0,
                pose_sigma_t=0.02, pose_sigma_deg=0.5, point_sigma=0.05, K=ICL_NUIM_K, w=640, h=480):
    """cfg4: points in a 4x3x3 m box 2-6 m in front of camera 0, cameras on a 1 m arc looking at the centroid, every
    point observed by every camera (or a random subset with `visibility` < 1), pixel noise + gross outliers, perturbed
    start, camera 0 fixed.  Returns a dict of arrays ready for BundleAdjustment / vs_ba_solve plus the ground truth."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K
    pts = np.empty((n_points, 3))
    pts[:, 0] = rng.uniform(-2.0, 2.0, n_points)
    pts[:, 1] = rng.uniform(-1.5, 1.5, n_points)
    pts[:, 2] = rng.uniform(2.5, 5.5, n_points)
    centroid = np.array([0.0, 0.0, 4.0])
    poses = np.zeros((n_cams, 4, 4))
    for i in range(n_cams):
        a = (i / max(n_cams - 1, 1)) * 1.0  # arc length 1 m
        pos = np.array([a, 0.05 * np.sin(3 * a), 0.1 * a * a])
        R = np.eye(3) if i == 0 else _look_at(pos, centroid)
        poses[i, :3, :3] = R
        poses[i, :3, 3] = pos
        poses[i, 3, 3] = 1.0
    obs_pose, obs_point, obs_uv = [], [], []
    for j in range(n_points):  # point-major, as the reference adds edges (LocalBA.py:164-172)
        for i in range(n_cams):
            if visibility < 1.0 and i > 1 and rng.random() > visibility:
                continue
            R, t = poses[i, :3, :3], poses[i, :3, 3]
            pc = R.T @ (pts[j] - t)
            if pc[2] <= 0.1:
                continue
            u, v = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
            obs_pose.append(i)
            obs_point.append(j)
            obs_uv.append((u, v))
    obs_pose = np.asarray(obs_pose, np.int32)
    obs_point = np.asarray(obs_point, np.int32)
    obs_uv_gt = np.asarray(obs_uv, np.float64)
    obs_uv = obs_uv_gt + rng.normal(0.0, noise_px, obs_uv_gt.shape) if noise_px > 0 else obs_uv_gt.copy()
    if outlier_frac > 0:
        bad = rng.random(obs_uv.shape[0]) < outlier_frac
        obs_uv[bad] += rng.uniform(-50, 50, (int(bad.sum()), 2))
    poses0 = poses.copy()
    for i in range(1, n_cams):
        if pose_sigma_t > 0:
            poses0[i, :3, 3] += rng.normal(0, pose_sigma_t, 3)
        if pose_sigma_deg > 0:
            ax = rng.normal(size=3)
            poses0[i, :3, :3] = poses0[i, :3, :3] @ _rot(ax, np.deg2rad(rng.normal(0, pose_sigma_deg)))
    pts0 = pts + (rng.normal(0, point_sigma, pts.shape) if point_sigma > 0 else 0.0)
    pose_fixed = np.zeros(n_cams, np.uint8)
    pose_fixed[0] = 1
    return dict(poses=poses0, pose_fixed=pose_fixed, points=pts0, point_fixed=np.zeros(n_points, np.uint8),
                obs_pose=obs_pose, obs_point=obs_point, obs_uv=obs_uv, K=K, poses_gt=poses, points_gt=pts,
                obs_uv_gt=obs_uv_gt)


def ba_sliding_window_workload(n_cams=100, n_points=200000, window=10, seed=3):
    """SURVEY 8d's scaled bundle adjustment: cameras on a straight 0.1 m-spaced track looking down +z, points 2.5-5.5 m
    ahead; point j is observed by `window` consecutive cameras starting at a random one -- the co-visibility key-frame BA
    has (a banded reduced camera system).  0.5 px noise, 2 % gross outliers, camera 0 fixed.  Observations grouped by point."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy = ICL_NUIM_K
    poses = np.tile(np.eye(4), (n_cams, 1, 1))
    poses[:, 0, 3] = 0.1 * np.arange(n_cams)
    start = r.integers(0, n_cams - window + 1, n_points)
    centre = 0.1 * (start + window / 2)
    pts = np.stack([centre + r.uniform(-1.0, 1.0, n_points), r.uniform(-1.2, 1.2, n_points), r.uniform(2.5, 5.5, n_points)], 1)
    cam = (start[:, None] + np.arange(window)[None, :]).astype(np.int32)            # [P, window]
    pt = np.repeat(np.arange(n_points, dtype=np.int32)[:, None], window, 1)
    pc = pts[pt.ravel()] - poses[cam.ravel(), :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1)
    uv += r.normal(0, 0.5, uv.shape)
    bad = r.random(len(uv)) < 0.02
    uv[bad] += r.uniform(-50, 50, (int(bad.sum()), 2))
    poses0 = poses.copy()
    poses0[1:, :3, 3] += r.normal(0, 0.01, (n_cams - 1, 3))
    pts0 = pts + r.normal(0, 0.03, pts.shape)
    fixed = np.zeros(n_cams, np.uint8)
    fixed[0] = 1
    return dict(poses=poses0, pose_fixed=fixed, points=pts0, point_fixed=np.zeros(n_points, np.uint8),
                obs_pose=cam.ravel(), obs_point=pt.ravel(), obs_uv=uv, K=ICL_NUIM_K, poses_gt=poses)
