"""ctypes front end of the CPU oracle (oracle/libvs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from
visual_slam_amd/.  Parity unpinned against cv2/g2o (see the header of oracle/vs_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_u8p = C.POINTER(C.c_uint8)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)


class BAProblem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("n_scale", C.c_int32),
        ("poses", c_f64p), ("pose_fixed", c_u8p), ("points", c_f64p), ("point_fixed", c_u8p),
        ("obs_pose", c_i32p), ("obs_point", c_i32p), ("obs_uv", c_f64p), ("obs_info", c_f64p),
        ("scale_parent", c_i32p), ("scale_child", c_i32p), ("scale_meas", c_f64p),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("huber_delta", C.c_double), ("dcs_phi", C.c_double),
        ("max_iterations", C.c_int32), ("reserved", C.c_int32),
    ]


class BAResult(C.Structure):
    _fields_ = [
        ("poses_out", c_f64p), ("points_out", c_f64p), ("chi2_trace", c_f64p), ("lambda_trace", c_f64p),
        ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
        ("iterations", C.c_int32), ("trials", C.c_int32), ("not_pd", C.c_int32), ("terminated", C.c_int32),
        ("trial_trace", c_f64p), ("trial_trace_cap", C.c_int32), ("reserved", C.c_int32),
    ]


def build(force=False, lib_path=None, extra_cflags=None):
    """Compile the oracle with gcc.  Returns the path of the shared library."""
    out = lib_path or os.path.join(_HERE, "libvs_oracle.so")
    src = os.path.join(_HERE, "vs_oracle.c")
    deps = [src, os.path.join(_HERE, "..", "include", "vslam_hip.h"),
            os.path.join(_HERE, "..", "include", "vs_brief_pattern.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cflags = ["-O3", "-march=x86-64-v2", "-mpopcnt"] if extra_cflags is None else list(extra_cflags)
    cmd = ["gcc", *cflags, "-fopenmp", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-std=c11", "-shared",
           "-o", out, src, "-lm"]
    subprocess.run(cmd, check=True, cwd=_HERE)
    return out


def load(lib_path=None):
    global _LIB
    if _LIB is not None and lib_path is None:
        return _LIB
    path = lib_path or os.path.join(_HERE, "libvs_oracle.so")
    if not os.path.exists(path):
        path = build(lib_path=lib_path)
    lib = C.CDLL(path)
    lib.vo_gray_mean3_u8.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, c_u8p]
    lib.vo_fast9_score_map.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_u8p]
    lib.vo_fast9_detect.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_u8p,
                                    C.POINTER(C.c_int)]
    lib.vo_boxsum5.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint16)]
    lib.vo_brief256.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_u8p, c_i32p, C.POINTER(C.c_int)]
    lib.vo_detect_describe_bgr.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_u8p, c_u8p,
                                           C.POINTER(C.c_int)]
    lib.vo_hamming_knn2.argtypes = [c_u8p, C.c_int, c_u8p, C.c_int, c_i32p, c_i32p]
    lib.vo_hamming_knn2_mt.argtypes = [c_u8p, C.c_int, c_u8p, C.c_int, c_i32p, c_i32p, C.c_int, C.POINTER(C.c_int)]
    lib.vo_match_ratio.argtypes = [c_u8p, C.c_int, c_u8p, C.c_int, C.c_double, c_i32p, c_i32p, c_i32p,
                                   C.POINTER(C.c_int)]
    lib.vo_cholesky_lower.argtypes = [c_f64p, C.c_int]
    lib.vo_ba_solve.argtypes = [C.POINTER(BAProblem), C.POINTER(BAResult)]
    lib.vo_sample_distinct.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, c_i32p]
    lib.vo_sample_distinct.restype = None
    lib.vo_eight_point.argtypes = [c_f64p, c_f64p, c_i32p, c_f64p]
    lib.vo_essential_ransac.argtypes = [c_f64p, c_f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64, c_f64p,
                                        c_u8p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    lib.vo_decompose_essential.argtypes = [c_f64p, c_f64p, c_f64p, c_f64p]
    lib.vo_decompose_essential.restype = None
    lib.vo_recover_pose.argtypes = [c_f64p, c_f64p, c_f64p, C.c_int, C.c_double, c_f64p, c_f64p, c_u8p, c_f64p,
                                    C.POINTER(C.c_int)]
    lib.vo_pnp_sample.argtypes = [C.c_uint64, C.c_int, C.c_int, c_i32p]
    lib.vo_pnp_sample.restype = None
    lib.vo_pnp_ransac.argtypes = [c_f64p, c_f64p, C.c_int, c_f64p, c_f64p, C.c_int, C.c_double, C.c_double, C.c_uint64,
                                  C.c_int, c_f64p, c_i32p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]
    lib.vo_ba_edge.argtypes = [c_f64p] * 7
    lib.vo_ba_pose_update.argtypes = [c_f64p] * 3
    if lib_path is None:
        _LIB = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _chk(rc, what):
    if rc != 0:
        raise ValueError("oracle %s failed with status %d" % (what, rc))


def _u8img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def gray_mean3(bgr, lib=None):
    lib = lib or load()
    bgr = _u8img(bgr)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    _chk(lib.vo_gray_mean3_u8(_p(bgr, c_u8p), w, h, 3 * w, _p(out, c_u8p)), "gray")
    return out


def fast9_score_map(gray, thr=20, border=3, lib=None):
    lib = lib or load()
    gray = _u8img(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    _chk(lib.vo_fast9_score_map(_p(gray, c_u8p), w, h, w, thr, border, _p(out, c_u8p)), "fast9_score_map")
    return out


def fast9_detect(gray, thr=20, border=3, max_kp=3000, lib=None):
    lib = lib or load()
    gray = _u8img(gray)
    h, w = gray.shape
    xy = np.zeros((max(max_kp, 1), 2), np.float32)
    sc = np.zeros(max(max_kp, 1), np.uint8)
    n = C.c_int(0)
    _chk(lib.vo_fast9_detect(_p(gray, c_u8p), w, h, w, thr, border, max_kp, _p(xy, c_f32p), _p(sc, c_u8p), C.byref(n)),
         "fast9_detect")
    return xy[:n.value].copy(), sc[:n.value].copy()


def boxsum5(gray, lib=None):
    lib = lib or load()
    gray = _u8img(gray)
    h, w = gray.shape
    out = np.empty((h, w), np.uint16)
    _chk(lib.vo_boxsum5(_p(gray, c_u8p), w, h, w, _p(out, C.POINTER(C.c_uint16))), "boxsum5")
    return out


def brief256(gray, xy, lib=None):
    lib = lib or load()
    gray = _u8img(gray)
    h, w = gray.shape
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = xy.shape[0]
    desc = np.zeros((max(n, 1), 32), np.uint8)
    keep = np.zeros(max(n, 1), np.int32)
    m = C.c_int(0)
    _chk(lib.vo_brief256(_p(gray, c_u8p), w, h, w, _p(xy, c_f32p), n, _p(desc, c_u8p), _p(keep, c_i32p), C.byref(m)),
         "brief256")
    return desc[:m.value].copy(), keep[:m.value].copy()


def detect_describe_bgr(bgr, thr=20, max_kp=3000, lib=None):
    lib = lib or load()
    bgr = _u8img(bgr)
    h, w, _ = bgr.shape
    xy = np.zeros((max(max_kp, 1), 2), np.float32)
    sc = np.zeros(max(max_kp, 1), np.uint8)
    desc = np.zeros((max(max_kp, 1), 32), np.uint8)
    n = C.c_int(0)
    _chk(lib.vo_detect_describe_bgr(_p(bgr, c_u8p), w, h, 3 * w, thr, max_kp, _p(xy, c_f32p), _p(sc, c_u8p),
                                    _p(desc, c_u8p), C.byref(n)), "detect_describe_bgr")
    return xy[:n.value].copy(), sc[:n.value].copy(), desc[:n.value].copy()


def hamming_knn2(q, t, threads=1, lib=None):
    lib = lib or load()
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.zeros((max(nq, 1), 2), np.int32)
    dist = np.zeros((max(nq, 1), 2), np.int32)
    used = C.c_int(0)
    _chk(lib.vo_hamming_knn2_mt(_p(q, c_u8p), nq, _p(t, c_u8p), nt, _p(idx, c_i32p), _p(dist, c_i32p), threads,
                                C.byref(used)), "hamming_knn2")
    return idx[:nq].copy(), dist[:nq].copy()


def match_ratio(q, t, ratio=0.8, lib=None):
    lib = lib or load()
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    nq, nt = q.shape[0], t.shape[0]
    mq = np.zeros(max(nq, 1), np.int32)
    mt = np.zeros(max(nq, 1), np.int32)
    md = np.zeros(max(nq, 1), np.int32)
    n = C.c_int(0)
    _chk(lib.vo_match_ratio(_p(q, c_u8p), nq, _p(t, c_u8p), nt, float(ratio), _p(mq, c_i32p), _p(mt, c_i32p),
                            _p(md, c_i32p), C.byref(n)), "match_ratio")
    return mq[:n.value].copy(), mt[:n.value].copy(), md[:n.value].copy()


def cholesky_lower(a, lib=None):
    """Returns (L, 0) or (garbage, k+1) when pivot k is not positive."""
    lib = lib or load()
    a = np.array(a, dtype=np.float64, order="C")
    n = a.shape[0]
    rc = lib.vo_cholesky_lower(_p(a, c_f64p), n)
    return np.tril(a), rc


def ba_solve(poses, pose_fixed, points, point_fixed, obs_pose, obs_point, obs_uv, K, huber_delta=np.sqrt(5.991),
             max_iterations=10, scale_edges=None, obs_info=None, dcs_phi=1.0, lib=None):
    """poses [F,4,4] camera-to-world; K = (fx, fy, cx, cy); scale_edges = (parent[], child[], meas[]) or None.
    Returns dict(poses, points, chi2_trace, lambda_trace, chi2_initial, chi2_final, lambda_final, iterations, trials,
    not_pd, terminated)."""
    lib = lib or load()
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
    point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
    obs_pose = np.ascontiguousarray(obs_pose, np.int32)
    obs_point = np.ascontiguousarray(obs_point, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
    p = BAProblem()
    p.n_poses, p.n_points, p.n_obs = poses.shape[0], points.shape[0], obs_pose.shape[0]
    p.poses, p.pose_fixed = _p(poses, c_f64p), _p(pose_fixed, c_u8p)
    p.points, p.point_fixed = _p(points, c_f64p), _p(point_fixed, c_u8p)
    p.obs_pose, p.obs_point, p.obs_uv = _p(obs_pose, c_i32p), _p(obs_point, c_i32p), _p(obs_uv, c_f64p)
    if obs_info is not None:
        obs_info = np.ascontiguousarray(obs_info, np.float64).reshape(-1, 3)
        p.obs_info = _p(obs_info, c_f64p)
    keep = []
    if scale_edges is not None and len(scale_edges[0]):
        sp = np.ascontiguousarray(scale_edges[0], np.int32)
        sc = np.ascontiguousarray(scale_edges[1], np.int32)
        sm = np.ascontiguousarray(scale_edges[2], np.float64)
        keep = [sp, sc, sm]
        p.n_scale = sp.shape[0]
        p.scale_parent, p.scale_child, p.scale_meas = _p(sp, c_i32p), _p(sc, c_i32p), _p(sm, c_f64p)
    p.fx, p.fy, p.cx, p.cy = (float(v) for v in K)
    p.huber_delta = float(huber_delta) if huber_delta else 0.0
    p.dcs_phi = float(dcs_phi)
    p.max_iterations = int(max_iterations)
    r = BAResult()
    poses_out = np.zeros_like(poses)
    points_out = np.zeros_like(points)
    chi = np.full(max(max_iterations, 1), np.nan)
    lam = np.full(max(max_iterations, 1), np.nan)
    r.poses_out, r.points_out = _p(poses_out, c_f64p), _p(points_out, c_f64p)
    r.chi2_trace, r.lambda_trace = _p(chi, c_f64p), _p(lam, c_f64p)
    tt = np.full((max(10 * max_iterations, 1), 4), np.nan)  # per-trial rows: lambda, trial chi2, rho, solve ok
    r.trial_trace, r.trial_trace_cap = _p(tt, c_f64p), tt.shape[0]
    _chk(lib.vo_ba_solve(C.byref(p), C.byref(r)), "ba_solve")
    del keep
    return dict(poses=poses_out.reshape(-1, 4, 4), points=points_out, chi2_trace=chi[:r.iterations].copy(),
                lambda_trace=lam[:r.iterations].copy(), chi2_initial=r.chi2_initial, chi2_final=r.chi2_final,
                lambda_final=r.lambda_final, iterations=r.iterations, trials=r.trials, not_pd=r.not_pd,
                terminated=r.terminated, trial_trace=tt[:min(r.trials, tt.shape[0])].copy())


def ba_edge(pose, X, K, uv, lib=None):
    lib = lib or load()
    pose = np.ascontiguousarray(pose, np.float64).reshape(16)
    X = np.ascontiguousarray(X, np.float64)
    K = np.ascontiguousarray(K, np.float64)
    uv = np.ascontiguousarray(uv, np.float64)
    e = np.zeros(2)
    Ji = np.zeros((2, 3))
    Jj = np.zeros((2, 6))
    _chk(lib.vo_ba_edge(_p(pose, c_f64p), _p(X, c_f64p), _p(K, c_f64p), _p(uv, c_f64p), _p(e, c_f64p), _p(Ji, c_f64p),
                        _p(Jj, c_f64p)), "ba_edge")
    return e, Ji, Jj


def ba_pose_update(pose, d6, lib=None):
    lib = lib or load()
    pose = np.ascontiguousarray(pose, np.float64).reshape(16)
    d6 = np.ascontiguousarray(d6, np.float64)
    out = np.zeros(16)
    _chk(lib.vo_ba_pose_update(_p(pose, c_f64p), _p(d6, c_f64p), _p(out, c_f64p)), "ba_pose_update")
    return out.reshape(4, 4)


def pnp_ransac(obj, img, K, pose0, iterations=100, reproj_err=8.0, confidence=0.99, seed=0, refine_iters=10, lib=None):
    """obj [N,3], img [N,2], pose0 = camera-to-world 4x4 guess.  Returns dict(found, pose, inliers, best_h, used)."""
    lib = lib or load()
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64)
    pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
    n = obj.shape[0]
    pose = np.zeros(16)
    inl = np.zeros(max(n, 1), np.int32)
    ni, found, bh, used = C.c_int(0), C.c_int(0), C.c_int(-1), C.c_int(0)
    _chk(lib.vo_pnp_ransac(_p(obj, c_f64p), _p(img, c_f64p), n, _p(K, c_f64p), _p(pose0, c_f64p), iterations,
                           reproj_err, confidence, seed, refine_iters, _p(pose, c_f64p), _p(inl, c_i32p), C.byref(ni),
                           C.byref(found), C.byref(bh), C.byref(used)), "pnp_ransac")
    return dict(found=bool(found.value), pose=pose.reshape(4, 4), inliers=inl[:ni.value].copy(), best_h=bh.value,
                used=used.value)


def pnp_sample(seed, h, n, lib=None):
    lib = lib or load()
    idx = np.zeros(5, np.int32)
    lib.vo_pnp_sample(seed, h, n, _p(idx, c_i32p))
    return idx


def eight_point(x1, x2, idx8, lib=None):
    """K-normalised correspondences + 8 indices -> essential matrix [3,3] (x2^T E x1 = 0) or None if degenerate."""
    lib = lib or load()
    x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(idx8, np.int32)
    E = np.zeros(9)
    ok = lib.vo_eight_point(_p(x1, c_f64p), _p(x2, c_f64p), _p(idx, c_i32p), _p(E, c_f64p))
    return E.reshape(3, 3) if ok else None


def essential_ransac(x1, x2, threshold, prob=0.999, max_iters=1000, seed=0, lib=None):
    """x1, x2 K-normalised [N,2] -> dict(found, E [3,3], mask uint8[N] (0/1), best_h, used)."""
    lib = lib or load()
    x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
    n = x1.shape[0]
    E = np.zeros(9)
    mask = np.zeros(max(n, 1), np.uint8)
    ni, found, bh, used = C.c_int(0), C.c_int(0), C.c_int(-1), C.c_int(0)
    _chk(lib.vo_essential_ransac(_p(x1, c_f64p), _p(x2, c_f64p), n, threshold, prob, max_iters, seed, _p(E, c_f64p),
                                 _p(mask, c_u8p), C.byref(ni), C.byref(found), C.byref(bh), C.byref(used)),
         "essential_ransac")
    return dict(found=bool(found.value), E=E.reshape(3, 3), mask=mask[:n].copy(), n_inliers=ni.value, best_h=bh.value,
                used=used.value)


def decompose_essential(E, lib=None):
    lib = lib or load()
    E = np.ascontiguousarray(E, np.float64).reshape(9)
    R1, R2, t = np.zeros(9), np.zeros(9), np.zeros(3)
    lib.vo_decompose_essential(_p(E, c_f64p), _p(R1, c_f64p), _p(R2, c_f64p), _p(t, c_f64p))
    return R1.reshape(3, 3), R2.reshape(3, 3), t


def recover_pose(E, x1, x2, dist_thresh=50.0, lib=None):
    """-> dict(R [3,3], t [3], mask uint8[N] (255/0), X [N,4] homogeneous, n_good)."""
    lib = lib or load()
    E = np.ascontiguousarray(E, np.float64).reshape(9)
    x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
    x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
    n = x1.shape[0]
    R, t = np.zeros(9), np.zeros(3)
    mask = np.zeros(max(n, 1), np.uint8)
    X = np.zeros((max(n, 1), 4))
    ng = C.c_int(0)
    _chk(lib.vo_recover_pose(_p(E, c_f64p), _p(x1, c_f64p), _p(x2, c_f64p), n, dist_thresh, _p(R, c_f64p), _p(t, c_f64p),
                             _p(mask, c_u8p), _p(X, c_f64p), C.byref(ng)), "recover_pose")
    return dict(R=R.reshape(3, 3), t=t, mask=mask[:n].copy(), X=X[:n].copy(), n_good=ng.value)
