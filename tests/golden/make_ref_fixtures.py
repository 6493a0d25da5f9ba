#!/usr/bin/env python3
"""Writes tests/golden/ref_fixtures.npz: values computed by the REFERENCE'S OWN SOURCE, executed in the build container.

Run in the build container only (it reads /root/reference; the GPU box has no reference):
    python tests/golden/make_ref_fixtures.py

What is executed.  The reference's modules cannot be imported (each imports cv2 and/or g2o at the top; neither exists in
this image), and nothing is installed or faked.  But a number of its functions and classes are pure Python /
NumPy.  Their source is taken out of the files with `ast` (the FunctionDef / ClassDef node, compiled as it stands) and run
in a namespace that holds only `np` and `deepcopy`:

  src/v2/helper_functions.py  triangulate (281-291), MakeHomogeneous (362-364), CameraProjectionMatrix (367-371),
                              CameraProjectionMatrix2 (376-377)
  src/v2/point.py             class Point (4-59)
  src/v2/map.py               class Map (6-131)      (visualize_map names g2o and is never called)
  src/v2/frame.py             class Frame (51-125)   (__init__ calls cv2.imread and is never called: instances are made with
                                                      object.__new__ and given the attributes __init__ assigns, 57-68)
                              FeatureMatcher.match_features (20-49): the Lowe-ratio loop and the gathers.  Its k-NN table
                                                      comes from `self.matcher.knnMatch`, a cv2 object in the reference; here
                                                      `self` is a plain object whose matcher returns the Hamming 2-NN table of
                                                      the NumPy twin (tests/np_twin.py) -- what is pinned is the ratio rule,
                                                      the gathers and the output order, not the k-NN search
  src/v2/LocalBA.py           BundleAdjustment.localBundleAdjustement (143-190), motionOnlyBundleAdjustement (195-229): the
                              P x F graph construction and the write-back, run with `self` = tests/ref_scenarios.py's
                              GraphRecorder (records add_pose / add_point / add_edge / AddScalingEdge; optimize() is the
                              identity) -- what is pinned is the problem handed to the solver (vertex order, fixed flags, edge
                              order, scale edges) and the write-back arithmetic (median normalisation), SURVEY 8a A14 / A15;
                              the solve itself lives in g2o and stays unpinned
  src/v1/testing.py           the hard-coded P1, P2, P3, x1 .. x3h2 (46-71): reference-held inputs, used for triangulate

Only DATA is written: inputs and the values the reference's code returned.  No reference source text is stored.
"""
import ast
import os
import sys
from copy import deepcopy

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VS_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import np_twin  # noqa: E402
import ref_scenarios as sc  # noqa: E402


def _tree(rel):
    path = os.path.join(REF, rel)
    return ast.parse(open(path).read(), filename=path), path


def extract(rel, name, inside=None):
    """The object the reference's source defines under `name` (a top-level def / class, or a method of class `inside`),
    from executing that one node."""
    tree, path = _tree(rel)
    body = tree.body
    if inside is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == inside).body
    node = next(n for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == name)
    if isinstance(node, ast.ClassDef):
        node.bases = []  # (Point, Map and Frame have none)
    ns = {"np": np, "deepcopy": deepcopy}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
    return ns[name]


def assigned(rel, names):
    """Module-level `NAME = <expression of np.array literals>` assignments of a reference file, evaluated."""
    tree, path = _tree(rel)
    out = {}
    for n in tree.body:
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and n.targets[0].id in names:
            out[n.targets[0].id] = eval(compile(ast.Expression(n.value), path, "eval"), {"np": np})
    return out


def main():
    out = {}
    # ---- pure-NumPy helpers ------------------------------------------------------------------------------------------------
    triangulate = extract("src/v2/helper_functions.py", "triangulate")
    make_h = extract("src/v2/helper_functions.py", "MakeHomogeneous")
    cpm = extract("src/v2/helper_functions.py", "CameraProjectionMatrix")
    cpm2 = extract("src/v2/helper_functions.py", "CameraProjectionMatrix2")
    held = assigned("src/v1/testing.py", {"P1", "P2", "P3", "x1", "x2", "x3", "x1h2", "x2h2", "x3h2"})
    assert set(held) >= {"P1", "P2", "P3", "x1", "x2", "x3"}
    for k, v in held.items():
        out["v1_" + k] = np.asarray(v, np.float64)
    # (a) the reference's hard-coded cameras and points, every pair of views
    for a, b in ((1, 2), (1, 3), (2, 3)):
        Pa, Pb = held["P%d" % a], held["P%d" % b]
        xa = np.vstack([held["x%d" % a], held["x%dh2" % a][:, :2]])
        xb = np.vstack([held["x%d" % b], held["x%dh2" % b][:, :2]])
        out["tri_v1_%d%d_pts1" % (a, b)], out["tri_v1_%d%d_pts2" % (a, b)] = xa, xb
        out["tri_v1_%d%d_X" % (a, b)] = triangulate(Pa, Pb, xa, xb)
    # (b) a synthetic two-view scene through the reference's own projection helpers
    rng = np.random.default_rng(21)
    K = np.array([[481.2, 0.0, 319.5], [0.0, 480.0, 239.5], [0.0, 0.0, 1.0]])
    n = 200
    X = rng.uniform(-1.5, 1.5, (n, 3)) + np.array([0.0, 0.0, 4.0])
    a = 0.1
    R = np.array([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])
    t = np.array([[0.3], [0.02], [0.05]])
    w2c1, w2c2 = np.eye(4), np.eye(4)
    w2c2[:3, :3], w2c2[:3, 3:] = R, t
    P1s, P2s = cpm2(w2c1, K), cpm2(w2c2, K)
    Xh = make_h(X)
    x1s = (P1s @ Xh.T).T
    x2s = (P2s @ Xh.T).T
    x1s = x1s[:, :2] / x1s[:, 2:] + rng.normal(0, 0.3, (n, 2))
    x2s = x2s[:, :2] / x2s[:, 2:] + rng.normal(0, 0.3, (n, 2))
    out.update(tri_syn_K=K, tri_syn_w2c1=w2c1, tri_syn_w2c2=w2c2, tri_syn_P1=P1s, tri_syn_P2=P2s, tri_syn_Xh=Xh,
               tri_syn_pts1=x1s, tri_syn_pts2=x2s, tri_syn_X=triangulate(P1s, P2s, x1s, x2s),
               cpm_R=R, cpm_t=t, cpm_out=cpm(R, t, K))
    # (c) the two-view scenes of tests/test_triangulate.py (sizes 1 .. 3 000, with and without pixel noise): the GPU tests
    #     compare vs_triangulate_dlt with THESE outputs of the reference's function, not with a restatement of it
    from visual_slam_amd.workloads import ba_workload
    for case, (n, noise, seed) in enumerate(sc.TWO_VIEW_CASES):
        w = ba_workload(n_cams=2, n_points=n, seed=seed, noise_px=noise, outlier_frac=0, pose_sigma_t=0, pose_sigma_deg=0,
                        point_sigma=0)
        p1, p2 = np.linalg.inv(w["poses_gt"][0]), np.linalg.inv(w["poses_gt"][1])
        uv = w["obs_uv"].reshape(n, 2, 2)
        Pa, Pb = cpm2(p1, K), cpm2(p2, K)
        xa, xb = make_h(uv[:, 0]), make_h(uv[:, 1])
        pre = "tri_tv%d_" % case
        out.update({pre + "w2c1": p1, pre + "w2c2": p2, pre + "P1": Pa, pre + "P2": Pb, pre + "x1": xa, pre + "x2": xb,
                    pre + "gt": w["points_gt"], pre + "X": triangulate(Pa, Pb, xa, xb)})
    # ---- Map / Point / Frame (host API mirror) -----------------------------------------------------------------------------
    Point = extract("src/v2/point.py", "Point")
    Map = extract("src/v2/map.py", "Map")
    Frame = extract("src/v2/frame.py", "Frame")

    def make_frame(i):
        f = object.__new__(Frame)  # Frame.__init__ would call cv2.imread: give the instance what it assigns instead
        f.rgb = f.depth = None
        f.keypoints, f.features = None, None
        f.ID = i
        f.pose = None
        f.parents = {}
        f.childs = []
        f.keyframe = False
        return f

    for k, v in sc.map_script(Map, Point, make_frame).items():
        out["map_" + k] = v
    # ---- graph construction + write-back of the two BA entry points ---------------------------------------------------------
    for i, (name, kw) in enumerate(sc.GRAPH_CASES):
        method = extract("src/v2/LocalBA.py", name, inside="BundleAdjustment")
        m, _, _ = sc.build_map(Map, Point, make_frame, seed=31 + i)
        rec = sc.GraphRecorder()
        method(rec, m, **kw)
        assert rec.optimized == 1
        for k, v in {**rec.arrays(), **sc.map_state(m)}.items():
            out["%s_%s" % (sc.graph_case_name(i), k)] = v
    # ---- Lowe ratio loop + gathers ---------------------------------------------------------------------------------------------
    match_features = extract("src/v2/frame.py", "match_features", inside="FeatureMatcher")

    class DM:
        def __init__(self, q, t, d):
            self.queryIdx, self.trainIdx, self.distance = int(q), int(t), float(d)

    class KnnTable:
        def __init__(self, idx, dist):
            self.idx, self.dist = idx, dist

        def knnMatch(self, desc1, desc2, k=2):
            assert k == 2
            return [(DM(q, self.idx[q, 0], self.dist[q, 0]), DM(q, self.idx[q, 1], self.dist[q, 1])) for q in range(len(desc1))]

    class Self:
        pass

    from visual_slam_amd.workloads import match_workload
    for case, (nq, nt, seed, ratio) in enumerate(((300, 400, 7, 0.8), (257, 129, 8, 0.7), (64, 2, 9, 0.95), (120, 500, 10, 1.0))):
        q, t = match_workload(nq, nt, n_dup=min(8, nt // 2), seed=seed)
        idx, dist = np_twin.hamming_knn2(q, t)
        rng = np.random.default_rng(100 + case)
        kp1 = rng.uniform(0, 640, (nq, 2)).astype(np.float32)
        kp2 = rng.uniform(0, 640, (nt, 2)).astype(np.float32)
        s = Self()
        s.matcher = KnnTable(idx, dist)
        matches, pts1, ft1, pts2, ft2 = match_features(s, kp1, q, kp2, t, ratio=ratio)
        pre = "mf%d_" % case
        out.update({pre + "kp1": kp1, pre + "desc1": q, pre + "kp2": kp2, pre + "desc2": t, pre + "ratio": np.float64(ratio),
                    pre + "knn_idx": idx.astype(np.int32), pre + "knn_dist": dist.astype(np.int32),
                    pre + "query": np.asarray([m[0].queryIdx for m in matches], np.int32),
                    pre + "train": np.asarray([m[0].trainIdx for m in matches], np.int32),
                    pre + "distance": np.asarray([m[0].distance for m in matches], np.float64),
                    pre + "rows_are_singletons": np.asarray([len(m) == 1 for m in matches]).all(),
                    pre + "pts1": np.asarray(pts1, np.float32).reshape(-1, 2), pre + "pts2": np.asarray(pts2, np.float32).reshape(-1, 2),
                    pre + "ft1": np.asarray(ft1, np.uint8).reshape(-1, 32), pre + "ft2": np.asarray(ft2, np.uint8).reshape(-1, 32)})
    path = os.path.join(HERE, "ref_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
