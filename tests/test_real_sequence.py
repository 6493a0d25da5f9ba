"""Real-sequence evidence (round-4 verdict, missing #4 / weak #4): the driver was run in the build container with the CPU-oracle back
ends over the first 420 frames of the reference's own data set with the reference's own key-frame rule (gap 20, min 80 tracked);
tests/golden/make_real_sequence.py stored the trajectory, the error against the ground truth and THE BUNDLE-ADJUSTMENT PROBLEMS THE
DRIVER ACTUALLY HANDED TO ITS SOLVER at three key frames (early / middle / last).

CPU: the stored record is self-consistent and says what the README says; the oracle reproduces its stored answers bit for bit.
GPU: the HIP solver on those real problems against the oracle, and WHICH kernels each problem takes -- real co-visibility (a point
seen from key frames up to 19 slots apart, a scale edge per key frame) is not the banded synthetic scene the large-window paths
were written against.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

NAMES = ("early", "middle", "last")
GUARDED = "guarded"   # the key-frame pose adjustment of the guarded run: every point fixed, 21 free cameras, a scale edge each
CONTRACT = 1e-4   # BASELINE.json: BA poses within 1e-4 relative Frobenius


def _load(name):
    f = np.load(os.path.join(GOLDEN, "real_ba_%s.npz" % name))
    d = {k: f[k] for k in f.files}
    args = (d["poses"], d["pose_fixed"], d["points"], d["point_fixed"], d["obs_pose"], d["obs_point"], d["obs_uv"], tuple(d["K"]))
    kw = dict(huber_delta=float(d["huber_delta"]), max_iterations=int(d["max_iterations"]), dcs_phi=float(d["dcs_phi"]),
              scale_edges=(d["scale_parent"].tolist(), d["scale_child"].tolist(), d["scale_meas"].tolist()))
    return d, args, kw


def _rel(a, b):
    return max(np.linalg.norm(x - y) / np.linalg.norm(y) for x, y in zip(a, b))


def _self_spread(oracle, d, kw, base, reps=4):
    """How far the oracle moves from itself when its edge list is merely reordered (another summation order of the same problem):
    the yardstick for an implementation with yet another order (tests/test_gpu_ba.py::_oracle_sensitivity).  These real
    problems are badly conditioned -- the map's scale has collapsed by then (README), lambda_0 = 1e-5 max diag H reaches 1e15."""
    rng = np.random.default_rng(7)
    poses = chi2 = 0.0
    for _ in range(reps):
        p = rng.permutation(len(d["obs_pose"]))
        o = oracle.ba_solve(d["poses"], d["pose_fixed"], d["points"], d["point_fixed"], d["obs_pose"][p], d["obs_point"][p],
                            d["obs_uv"][p], tuple(d["K"]), **kw)
        poses = max(poses, _rel(o["poses"], base["poses"]))
        chi2 = max(chi2, abs(o["chi2_final"] - base["chi2_final"]) / base["chi2_final"])
    return poses, chi2


def test_stored_record_is_consistent():
    s = json.load(open(os.path.join(GOLDEN, "real_sequence.json")))
    z = np.load(os.path.join(GOLDEN, "real_sequence.npz"))
    from visual_slam_amd import dataset
    assert s["frames"] >= 400 and s["keyframe_gap"] == 20 and s["min_tracked"] == 80
    for init in ("depth", "two_view", "depth_guarded", "two_view_guarded"):
        r = s["runs"][init]
        P = z["poses_" + init]
        assert P.shape == (s["frames"], 4, 4) and z["gt"].shape == P.shape
        ate = dataset.ate_rmse(P, z["gt"])
        assert abs(ate["rmse"] - r["ate_rmse_m"]) < 1e-9 and r["gt_path_length_m"] > 2.0      # a path of metres
        assert len(r["keyframes"]) >= 15 and r["keyframes"][0] == 0
        gaps = np.diff(r["keyframes"][1:] if init.startswith("two_view") else r["keyframes"])
        assert gaps.max() <= 21                                                             # main.py:221: i - loop_idx > 20
        assert len(r["local_ba"]) == len(r["keyframes"]) - 1 and r["local_ba"][-1]["poses"] == len(r["keyframes"])
        for P_ in P:
            assert np.allclose(P_[:3, :3] @ P_[:3, :3].T, np.eye(3), atol=1e-8)
    # what the record is there to say: main.py's control flow loses the scale (half a metre of error on 2.5 m), the same
    # kernels behind two guards track the same 420 frames to centimetres
    assert s["runs"]["depth"]["ate_rmse_m"] > 0.4 and s["runs"]["two_view"]["ate_rmse_m"] > 0.4
    g = s["runs"]["depth_guarded"]
    assert g["ate_rmse_m"] < 0.10 and g["ate_rmse_first_half_m"] < 0.03 and g["ate_rmse_first_350_frames_m"] < 0.04
    assert 3.0 < g["sim3_scale"] < 4.0                     # the map's unit stays the initialisation's (median point norm 3.41 m)
    assert s["runs"]["two_view_guarded"]["ate_rmse_m"] < 0.15
    for name in NAMES + (GUARDED,):
        d, _, _ = _load(name)
        fx = s["ba_fixtures"][name]
        assert (len(d["poses"]), len(d["points"]), len(d["obs_pose"])) == (fx["poses"], fx["points"], fx["observations"])
        assert d["pose_fixed"].tolist() == [1] + [0] * (len(d["poses"]) - 1)                    # LocalBA.py:155-156
        assert len(d["scale_parent"]) == len(d["poses"]) - 1                                    # LocalBA.py:159-162
        assert bool(d["point_fixed"].all()) == (name == GUARDED) and (name == GUARDED or not d["point_fixed"].any())  # LocalBA.py:165 / the guard
        assert np.all(np.diff(d["obs_point"]) >= 0)                                            # point-major (LocalBA.py:164-172)


@pytest.mark.parametrize("name", NAMES + (GUARDED,))
def test_oracle_reproduces_its_stored_answers(oracle, name):
    d, args, kw = _load(name)
    o = oracle.ba_solve(*args, **kw)
    assert np.array_equal(o["poses"], d["oracle_poses"]) and o["chi2_final"] == float(d["oracle_chi2_final"])
    assert o["chi2_initial"] == float(d["oracle_chi2_initial"]) and o["trials"] == int(d["oracle_trials"])


# kernels each real problem must take (Context.ba_last_path): the free cameras are poses - 1
EXPECT = {
    "early": dict(schur="ba_schur_small", dense="ba_solve_block"),     # 4 free cameras: one tile
    # 11 free cameras = 2 tiles, every point's cameras within 16 slots: the banded-window Schur kernel -- but the scale edges
    # couple camera pairs in Hpp, so the factorisation is the dense one (band = 0), in LDS (66 unknowns)
    "middle": dict(schur="ba_schur_window", dense="ba_solve_block", band=0),
    # 21 free cameras, points seen from key frames up to 19 slots apart: OFF the banded window (> 16 slots) -> the tile kernel;
    # 126 unknowns: the last size ba_solve_block takes
    "last": dict(schur="ba_schur_tile", dense="ba_solve_block", band=0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_real_problems_on_the_hip_solver_equal_the_oracle(vs, oracle, name):
    d, args, kw = _load(name)
    o = oracle.ba_solve(*args, **kw)
    g = vs.ba_solve(*args, **kw)
    path = vs.ba_last_path()
    for k, v in EXPECT[name].items():
        assert path[k] == v, (name, path)
    assert path["unknowns"] == 6 * (len(d["poses"]) - 1)
    sp_pose, sp_chi2 = _self_spread(oracle, d, kw, o)
    rel = _rel(g["poses"], o["poses"])
    print("real_ba_%s: %s; HIP vs oracle poses %.2e (oracle vs itself with reordered edges %.2e), chi2 %.6e vs %.6e" % (
        name, path, rel, sp_pose, g["chi2_final"], o["chi2_final"]))
    assert rel <= CONTRACT
    assert rel <= max(1e-8, 20.0 * sp_pose), (rel, sp_pose)
    assert np.isclose(g["chi2_initial"], o["chi2_initial"], rtol=1e-10)
    assert np.isclose(g["chi2_final"], o["chi2_final"], rtol=max(1e-9, 20.0 * sp_chi2))
    assert g["iterations"] == o["iterations"] and g["trials"] == o["trials"] and g["not_pd"] == o["not_pd"]
    # the first trial starts from identical states
    assert np.isclose(g["chi2_trace"][0], o["chi2_trace"][0], rtol=max(1e-9, 20.0 * sp_chi2))
    assert np.isclose(g["lambda_trace"][0], o["lambda_trace"][0], rtol=1e-9)


@pytest.mark.gpu
def test_guarded_keyframe_pose_adjustment_on_the_hip_solver(vs, oracle):
    """The guarded driver's key-frame step (BundleAdjustment.keyframePoseAdjustement: every point FIXED, the key-frame poses free,
    a scale edge per key frame) on a real problem: 21 free cameras, 4 715 observations.  No free point, so there is no Schur
    complement -- but the scale edges couple the cameras, so it is NOT the block-diagonal motion-only problem either."""
    d, args, kw = _load(GUARDED)
    o = oracle.ba_solve(*args, **kw)
    g = vs.ba_solve(*args, **kw)
    path = vs.ba_last_path()
    print("real_ba_guarded: %s" % path)
    assert path["unknowns"] == 126 and path["schur"] != "none (motion-only)", path
    sp_pose, sp_chi2 = _self_spread(oracle, d, kw, o)
    rel = _rel(g["poses"], o["poses"])
    print("real_ba_guarded: HIP vs oracle poses %.2e (oracle vs itself %.2e), chi2 %.6e vs %.6e, trials %d / %d" % (
        rel, sp_pose, g["chi2_final"], o["chi2_final"], g["trials"], o["trials"]))
    assert rel <= CONTRACT and rel <= max(1e-8, 20.0 * sp_pose)
    assert np.array_equal(g["points"], d["points"])                       # fixed points come back untouched
    assert np.isclose(g["chi2_initial"], o["chi2_initial"], rtol=1e-10)
    assert np.isclose(g["chi2_final"], o["chi2_final"], rtol=max(1e-9, 20.0 * sp_chi2))
    # With the points fixed this solve CONVERGES (chi2 stops moving in the ninth digit after a few iterations); from there on the
    # gain ratio is rounding noise and accept / reject decisions -- hence trial and iteration counts -- may differ between two
    # summation orders (tests/test_gpu_ba.py::_compare): lock step is asserted while the oracle's chi2 still moves
    tr = np.concatenate([[o["chi2_initial"]], o["chi2_trace"]])
    live = np.nonzero(np.abs(np.diff(tr)) > 1e-9 * tr[1:])[0]
    n_live = int(live[-1]) + 1 if len(live) else 0
    assert n_live >= 3 and np.allclose(g["chi2_trace"][:n_live], o["chi2_trace"][:n_live], rtol=max(1e-7, 20.0 * sp_chi2))
    assert np.allclose(g["lambda_trace"][:max(n_live - 1, 0)], o["lambda_trace"][:max(n_live - 1, 0)], rtol=1e-6)
    if n_live == len(o["chi2_trace"]):
        assert g["iterations"] == o["iterations"] and g["trials"] == o["trials"]


@pytest.mark.gpu
def test_real_problem_beyond_the_lds_solver(vs, oracle):
    """One key frame more than the stored run reached: 22 free cameras = 132 unknowns, past ba_solve_block's square layout -- with
    scale edges (no band) that was the blocked HBM factorisation, ba_chol_panel / ba_chol_update, and is ba_solve_block with the
    lower triangle packed since the second session of round 5 (up to 198 unknowns); both are run, both against the oracle, and they
    must agree with each other to rounding.  Built from the last real problem by appending a copy of its last key frame (the same points observed
    a quarter of a pixel away, the pose nudged, one more scale edge)."""
    d, args, kw = _load("last")
    n = len(d["poses"])
    nudge = np.eye(4)
    nudge[:3, 3] = [1e-4, -2e-4, 1e-4]
    poses = np.concatenate([d["poses"], (d["poses"][-1] @ nudge)[None]])
    fixed = np.r_[d["pose_fixed"], 0].astype(np.uint8)
    last = d["obs_pose"] == n - 1
    # keep the list grouped by point: the copy's observation goes right behind the original's
    order = np.argsort(np.r_[np.arange(len(last)), np.nonzero(last)[0] + 0.5], kind="stable")
    obs_pose = np.r_[d["obs_pose"], np.full(last.sum(), n, np.int32)][order].astype(np.int32)
    obs_point = np.r_[d["obs_point"], d["obs_point"][last]][order].astype(np.int32)
    obs_uv = np.concatenate([d["obs_uv"], d["obs_uv"][last] + 0.25])[order]
    se = (kw["scale_edges"][0] + [n - 1], kw["scale_edges"][1] + [n], kw["scale_edges"][2] + [2e-4])
    kw = dict(kw, scale_edges=se)
    a = (poses, fixed, d["points"], d["point_fixed"], obs_pose, obs_point, obs_uv, tuple(d["K"]))
    o = oracle.ba_solve(*a, **kw)
    g = vs.ba_solve(*a, **kw)
    path = vs.ba_last_path()
    assert path["unknowns"] == 132 and path["dense"] == "ba_solve_block" and path["schur"] == "ba_schur_tile", path
    try:
        vs.tune_ba_solve(1)
        g_hbm = vs.ba_solve(*a, **kw)
        path_hbm = vs.ba_last_path()
    finally:
        vs.tune_ba_solve(0)
    assert path_hbm["dense"] == "ba_chol_panel", path_hbm
    dd = dict(d, poses=poses, pose_fixed=fixed, obs_pose=obs_pose, obs_point=obs_point, obs_uv=obs_uv)
    sp_pose, sp_chi2 = _self_spread(oracle, dd, kw, o)
    rel = _rel(g["poses"], o["poses"])
    print("real problem + 1 key frame: %s; HIP vs oracle %.2e (oracle self-spread %.2e)" % (path, rel, sp_pose))
    assert rel <= CONTRACT and rel <= max(1e-8, 20.0 * sp_pose)
    assert np.isclose(g["chi2_final"], o["chi2_final"], rtol=max(1e-9, 20.0 * sp_chi2)) and g["trials"] == o["trials"]
    # the two dense paths: the same solution of the reduced system bit for bit (test_packed_lds_solver_is_bitwise_the_other_dense_paths),
    # but the camera update behind it normalises the quaternion with different code in the two kernels -- on a problem this badly
    # conditioned one ulp there is 2e-8 in the result: bounded like the distance to the oracle, by the oracle's own spread
    rel_hbm = _rel(g_hbm["poses"], o["poses"])
    assert rel_hbm <= CONTRACT and rel_hbm <= max(1e-8, 20.0 * sp_pose) and g_hbm["trials"] == o["trials"]
    assert _rel(g["poses"], g_hbm["poses"]) <= max(1e-8, 20.0 * sp_pose)
