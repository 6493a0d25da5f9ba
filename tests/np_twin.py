"""Independent NumPy restatement of the hot-path definitions, used ONLY to cross-check the C oracle on small cases.

Written from the definitions (not from oracle/vs_oracle.c) and deliberately with different algorithms:
  FAST   : 16-bit brighter/darker masks + "9 contiguous ones in a circular word" bit trick, score by threshold search
  BRIEF  : vectorised gather on an integral-image box sum
  Hamming: unpackbits + matrix arithmetic, stable argsort for the tie rule
  BA     : dense (no Schur) LM with central-difference Jacobians through the actual update rule
"""
import numpy as np

CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
          (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def gray_mean3(bgr):
    return np.mean(bgr, axis=2).astype(np.uint8)  # the reference's expression, src/v2/frame.py:11


def _has_run9(mask16):
    m = mask16.astype(np.uint32)
    m = m | (m << 16)
    r = m & (m >> 1)
    r = r & (r >> 2)
    r = r & (r >> 4)
    r = r & (m >> 8)
    return (r & 0xFFFF) != 0


def _is_corner(gray, t):
    g = gray.astype(np.int32)
    h, w = g.shape
    c = g[3:h - 3, 3:w - 3]
    bright = np.zeros(c.shape, np.uint32)
    dark = np.zeros(c.shape, np.uint32)
    for k, (dx, dy) in enumerate(CIRCLE):
        p = g[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
        bright |= ((p > c + t).astype(np.uint32) << k)
        dark |= ((p < c - t).astype(np.uint32) << k)
    out = np.zeros((h, w), bool)
    out[3:h - 3, 3:w - 3] = _has_run9(bright) | _has_run9(dark)
    return out


def fast9_score_map(gray, thr=20, border=3):
    """score = largest t at which the pixel is still a FAST-9 corner (0 if not a corner at thr)."""
    h, w = gray.shape
    score = np.zeros((h, w), np.int32)
    alive = _is_corner(gray, thr)
    t = thr
    while alive.any() and t <= 254:
        score[alive] = t
        t += 1
        alive = alive & _is_corner(gray, t)
    inside = np.zeros((h, w), bool)
    inside[border:h - border, border:w - border] = True
    score[~inside] = 0
    return score.astype(np.uint8)


def fast9_detect(gray, thr=20, border=3, max_kp=3000):
    s = fast9_score_map(gray, thr, border).astype(np.int32)
    h, w = s.shape
    pad = np.zeros((h + 2, w + 2), np.int32)
    pad[1:-1, 1:-1] = s
    nb = np.full((h, w), -1, np.int32)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                nb = np.maximum(nb, pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w])
    keep = (s > 0) & (s > nb)
    ys, xs = np.nonzero(keep)  # row-major
    sc = s[ys, xs]
    if len(sc) > max_kp:
        order = np.lexsort((ys * w + xs, -sc))[:max_kp]  # by score desc, then index asc
        order = np.sort(order)
        ys, xs, sc = ys[order], xs[order], sc[order]
    return np.stack([xs, ys], 1).astype(np.float32), sc.astype(np.uint8)


def boxsum5(gray):
    g = gray.astype(np.int64)
    h, w = g.shape
    ii = np.zeros((h + 1, w + 1), np.int64)
    ii[1:, 1:] = g.cumsum(0).cumsum(1)
    out = np.zeros((h, w), np.int64)
    out[2:h - 2, 2:w - 2] = ii[5:, 5:] - ii[:-5, 5:] - ii[5:, :-5] + ii[:-5, :-5]
    return out.astype(np.uint16)


def brief256(gray, xy, pattern):
    h, w = gray.shape
    box = boxsum5(gray).astype(np.int32)
    xy = np.asarray(xy, np.float32).reshape(-1, 2)
    xi = np.rint(xy[:, 0]).astype(np.int64)  # rint = half to even
    yi = np.rint(xy[:, 1]).astype(np.int64)
    ok = (xi >= 15) & (yi >= 15) & (xi < w - 15) & (yi < h - 15)
    keep = np.nonzero(ok)[0].astype(np.int32)
    xi, yi = xi[ok], yi[ok]
    pat = np.asarray(pattern, np.int64)
    a = box[yi[:, None] + pat[None, :, 1], xi[:, None] + pat[None, :, 0]]
    b = box[yi[:, None] + pat[None, :, 3], xi[:, None] + pat[None, :, 2]]
    bits = (a < b).astype(np.uint8)
    return np.packbits(bits, axis=1, bitorder="little"), keep


def hamming_knn2(q, t):
    qb = np.unpackbits(np.asarray(q, np.uint8), axis=1).astype(np.int32)
    tb = np.unpackbits(np.asarray(t, np.uint8), axis=1).astype(np.int32)
    d = qb.sum(1)[:, None] + tb.sum(1)[None, :] - 2 * (qb @ tb.T)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]  # stable: ties keep the lower index first
    return order.astype(np.int32), np.take_along_axis(d, order, 1).astype(np.int32)


def match_ratio(q, t, ratio=0.8):
    idx, dist = hamming_knn2(q, t)
    keep = dist[:, 0] < ratio * dist[:, 1]
    qi = np.nonzero(keep)[0].astype(np.int32)
    return qi, idx[keep, 0], dist[keep, 0]


# ------------------------------------------------------------------------------------------------------ BA twin
def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _R_quat(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()  # x y z w
    return -q if q[3] < 0 else q


def _cam_update(t, q, d):
    t = t + d[:3]
    v = d[3:]
    dq = np.array([v[0], v[1], v[2], np.sqrt(1.0 - v @ v)])
    q = _quat_mul(q, dq)
    return t, q / np.linalg.norm(q)


def _residuals(ts, qs, pts, obs_pose, obs_point, obs_uv, K):
    fx, fy, cx, cy = K
    e = np.zeros((len(obs_pose), 2))
    for o, (i, j) in enumerate(zip(obs_pose, obs_point)):
        pc = _quat_R(qs[i]).T @ (pts[j] - ts[i])
        e[o] = (fx * pc[0] / pc[2] + cx - obs_uv[o, 0], fy * pc[1] / pc[2] + cy - obs_uv[o, 1])
    return e


def _huber(e2, delta):
    if delta <= 0:
        return e2, np.ones_like(e2)
    s = np.sqrt(np.maximum(e2, 1e-300))
    inl = e2 <= delta * delta
    return np.where(inl, e2, 2 * s * delta - delta * delta), np.where(inl, 1.0, delta / s)


def ba_lm_dense(poses, pose_fixed, points, point_fixed, obs_pose, obs_point, obs_uv, K, huber_delta, max_iterations=10):
    """Dense LM (full normal equations, no Schur), numeric Jacobians through the real update rule.  No scale edges."""
    F, P = len(poses), len(points)
    ts = [poses[i][:3, 3].copy() for i in range(F)]
    qs = [_R_quat(poses[i][:3, :3]) for i in range(F)]
    pts = np.array(points, float)
    pslot = -np.ones(F, int)
    lslot = -np.ones(P, int)
    pslot[np.asarray(pose_fixed) == 0] = np.arange(int((np.asarray(pose_fixed) == 0).sum()))
    lslot[np.asarray(point_fixed) == 0] = np.arange(int((np.asarray(point_fixed) == 0).sum()))
    npz = 6 * int((pslot >= 0).sum())
    nx = npz + 3 * int((lslot >= 0).sum())
    active = np.array([(pslot[i] >= 0) or (lslot[j] >= 0) for i, j in zip(obs_pose, obs_point)])

    def chi2(ts_, qs_, pts_):
        e = _residuals(ts_, qs_, pts_, obs_pose, obs_point, obs_uv, K)
        rho, _ = _huber((e * e).sum(1), huber_delta)
        return rho[active].sum()

    def apply(ts_, qs_, pts_, x):
        ts2, qs2, pts2 = [t.copy() for t in ts_], [q.copy() for q in qs_], pts_.copy()
        for i in range(F):
            if pslot[i] >= 0:
                ts2[i], qs2[i] = _cam_update(ts_[i], qs_[i], x[6 * pslot[i]:6 * pslot[i] + 6])
        for j in range(P):
            if lslot[j] >= 0:
                pts2[j] = pts_[j] + x[npz + 3 * lslot[j]:npz + 3 * lslot[j] + 3]
        return ts2, qs2, pts2

    lam, ni = 0.0, 2.0
    trace = []
    for it in range(max_iterations):
        cur = chi2(ts, qs, pts)
        e0 = _residuals(ts, qs, pts, obs_pose, obs_point, obs_uv, K)
        _, w = _huber((e0 * e0).sum(1), huber_delta)
        H = np.zeros((nx, nx))
        b = np.zeros(nx)
        h = 1e-6
        for o, (i, j) in enumerate(zip(obs_pose, obs_point)):
            if not active[o]:
                continue
            cols, J = [], []
            if pslot[i] >= 0:
                for d in range(6):
                    dv = np.zeros(6)
                    dv[d] = h
                    tp, qp = _cam_update(ts[i], qs[i], dv)
                    tm, qm = _cam_update(ts[i], qs[i], -dv)
                    ep = _residuals([tp], [qp], pts[j:j + 1], [0], [0], obs_uv[o:o + 1], K)[0]
                    em = _residuals([tm], [qm], pts[j:j + 1], [0], [0], obs_uv[o:o + 1], K)[0]
                    J.append((ep - em) / (2 * h))
                    cols.append(6 * pslot[i] + d)
            if lslot[j] >= 0:
                for d in range(3):
                    dp = np.zeros(3)
                    dp[d] = h
                    ep = _residuals([ts[i]], [qs[i]], [pts[j] + dp], [0], [0], obs_uv[o:o + 1], K)[0]
                    em = _residuals([ts[i]], [qs[i]], [pts[j] - dp], [0], [0], obs_uv[o:o + 1], K)[0]
                    J.append((ep - em) / (2 * h))
                    cols.append(npz + 3 * lslot[j] + d)
            J = np.array(J).T  # 2 x k
            cols = np.array(cols)
            H[np.ix_(cols, cols)] += w[o] * (J.T @ J)
            b[cols] += -w[o] * (J.T @ e0[o])
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(H)).max()
        rho, q = 0.0, 0
        while True:
            try:
                L = np.linalg.cholesky(H + lam * np.eye(nx))
                x = np.linalg.solve(L.T, np.linalg.solve(L, b))
                ok = True
            except np.linalg.LinAlgError:
                x = np.zeros(nx)
                ok = False
            ts2, qs2, pts2 = apply(ts, qs, pts, x)
            tmp = chi2(ts2, qs2, pts2) if ok else np.finfo(float).max
            rho = (cur - tmp) / (x @ (lam * x + b) + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                ni = 2.0
                cur = tmp
                ts, qs, pts = ts2, qs2, pts2
            else:
                lam *= ni
                ni *= 2
            q += 1
            if not (rho < 0 and q < 10):
                break
        trace.append(cur)
        if q == 10 or rho == 0:
            break
    out = np.zeros((F, 4, 4))
    for i in range(F):
        out[i, :3, :3] = _quat_R(qs[i])
        out[i, :3, 3] = ts[i]
        out[i, 3, 3] = 1
    return out, pts, np.array(trace)
