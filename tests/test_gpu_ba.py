"""GPU parity: HIP bundle adjustment (LM + Schur, FP64) vs the CPU oracle through the C ABI.

Contract (BASELINE.json): every pose within 1e-4 relative Frobenius norm of the reference optimizer.  The kernels
follow the oracle's arithmetic order, so the tests also assert a much tighter agreement (1e-8) to catch logic errors
that the loose contract would hide.
"""
import numpy as np
import pytest

from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload

pytestmark = pytest.mark.gpu
HUBER = float(np.sqrt(5.991))
CONTRACT = 1e-4
TIGHT = 1e-8


def _args(w):
    return (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])


def _compare(g, o, tight=TIGHT):
    rel = [np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], o["poses"])]
    assert max(rel) <= CONTRACT, max(rel)          # the contract
    assert max(rel) <= tight, max(rel)             # what the implementation actually achieves
    assert np.allclose(g["points"], o["points"], rtol=0, atol=max(tight, 1e-12) * 10)
    assert np.isclose(g["chi2_initial"], o["chi2_initial"], rtol=1e-10)
    assert np.isclose(g["chi2_final"], o["chi2_final"], rtol=1e-9)
    # Once LM has converged to machine precision the gain ratio is rounding noise, so accept/reject decisions (and with
    # them the trial count and lambda) may legitimately differ; before that point the two runs must walk in lock step.
    tr = np.concatenate([[o["chi2_initial"]], o["chi2_trace"]])
    live = np.nonzero(np.abs(np.diff(tr)) > 1e-9 * tr[1:])[0]
    n_live = int(live[-1]) + 1 if len(live) else 0
    assert np.allclose(g["chi2_trace"][:n_live], o["chi2_trace"][:n_live], rtol=1e-7)
    assert np.allclose(g["lambda_trace"][:max(n_live - 1, 0)], o["lambda_trace"][:max(n_live - 1, 0)], rtol=1e-6)
    if n_live == len(o["chi2_trace"]):
        assert g["iterations"] == o["iterations"] and g["trials"] == o["trials"]
        assert g["terminated"] == o["terminated"] and g["not_pd"] == o["not_pd"]
    return max(rel)


def test_cfg4_10_cameras_2000_points(vs, oracle):
    w = ba_workload()  # BASELINE.json configs[3]
    g = vs.ba_solve(*_args(w), huber_delta=HUBER, max_iterations=10)
    o = oracle.ba_solve(*_args(w), huber_delta=HUBER, max_iterations=10)
    worst = _compare(g, o)
    assert g["iterations"] == 10 and g["chi2_final"] < 0.15 * g["chi2_initial"]
    print("cfg4 worst relative pose difference vs oracle: %.3e" % worst)


@pytest.mark.parametrize("n_cams,n_points,vis,seed", [(3, 20, 1.0, 5), (2, 300, 1.0, 6), (6, 150, 0.6, 13), (15, 400, 0.5, 21),
                                                    (5, 60, 1.0, 7)])
def test_small_scenes(vs, oracle, n_cams, n_points, vis, seed):
    w = ba_workload(n_cams=n_cams, n_points=n_points, seed=seed, visibility=vis)
    _compare(vs.ba_solve(*_args(w)), oracle.ba_solve(*_args(w)))


def test_no_robust_kernel_and_information(vs, oracle):
    w = ba_workload(n_cams=4, n_points=100, seed=31, outlier_frac=0)
    _compare(vs.ba_solve(*_args(w), huber_delta=0), oracle.ba_solve(*_args(w), huber_delta=0))
    rng = np.random.default_rng(1)
    n = len(w["obs_pose"])
    a = rng.uniform(0.5, 2.0, n)
    c = rng.uniform(0.5, 2.0, n)
    b = rng.uniform(-0.3, 0.3, n)
    info = np.stack([a, b, c], 1)
    _compare(vs.ba_solve(*_args(w), obs_info=info), oracle.ba_solve(*_args(w), obs_info=info))


def test_motion_only(vs, oracle):
    w = ba_workload(n_cams=8, n_points=500, seed=9, point_sigma=0, visibility=0.7)
    w["point_fixed"][:] = 1
    w["pose_fixed"][[0, 3]] = 1
    g = vs.ba_solve(*_args(w))
    o = oracle.ba_solve(*_args(w))
    _compare(g, o)
    assert np.array_equal(g["points"], w["points"]) and np.allclose(g["poses"][3], w["poses"][3], atol=1e-15)


def test_scale_edges(vs, oracle):
    w = ba_workload(n_cams=5, n_points=80, seed=4, noise_px=0.2, outlier_frac=0)
    meas = [np.linalg.norm(w["poses_gt"][i][:3, 3] - w["poses_gt"][i - 1][:3, 3]) * (1.0 + 0.05 * i) for i in range(1, 5)]
    se = ([0, 1, 2, 3], [1, 2, 3, 4], meas)
    g = vs.ba_solve(*_args(w), scale_edges=se)
    o = oracle.ba_solve(*_args(w), scale_edges=se)
    _compare(g, o)
    g0 = vs.ba_solve(*_args(w))
    assert not np.allclose(g["poses"], g0["poses"], atol=1e-9)


def test_mixed_fixed_points_and_unobserved_points(vs, oracle):
    w = ba_workload(n_cams=5, n_points=120, seed=17, visibility=0.8)
    w["point_fixed"][::3] = 1
    keep = w["obs_point"] % 10 != 7  # points 7, 17, ... lose all observations
    for k in ("obs_pose", "obs_point", "obs_uv"):
        w[k] = w[k][keep]
    _compare(vs.ba_solve(*_args(w)), oracle.ba_solve(*_args(w)))


def test_free_points_seen_by_fixed_cameras_only(vs, oracle):
    """Key-frame BA fixes frame 0 (LocalBA.py:155-156): points observed by fixed cameras alone are still free points --
    they get the damped 3x3 update but touch no block of the reduced camera system."""
    w = ba_workload(n_cams=4, n_points=90, seed=29)
    w["pose_fixed"][:2] = 1                                       # cameras 0 and 1 fixed
    only_fixed = (w["obs_point"] % 3 == 0) & (w["obs_pose"] >= 2)  # every third point loses its free-camera observations
    for k in ("obs_pose", "obs_point", "obs_uv"):
        w[k] = w[k][~only_fixed]
    g, o = vs.ba_solve(*_args(w)), oracle.ba_solve(*_args(w))
    _compare(g, o)
    moved = np.linalg.norm(g["points"][::3] - w["points"][::3], axis=1)
    assert moved.min() > 1e-6                                      # those points were optimised, not skipped


def test_duplicate_observations_are_ordered_and_deterministic(vs, oracle):
    """One camera observing a point twice (never produced by the reference's dict of frames, legal at the ABI): several
    (observation, observation) pairs hit one element of the Schur slab.  They are accumulated in ordered rounds (no
    atomics), so the result meets the tight bound and is bit-identical from run to run."""
    w = ba_workload(n_cams=3, n_points=30, seed=23)
    for k in ("obs_pose", "obs_point"):
        w[k] = np.concatenate([w[k], w[k][:12], w[k][:5]])   # 12 points seen twice, 5 of them three times
    w["obs_uv"] = np.concatenate([w["obs_uv"], w["obs_uv"][:12] + 0.3, w["obs_uv"][:5] - 0.2])
    a = vs.ba_solve(*_args(w))
    _compare(a, oracle.ba_solve(*_args(w)), tight=TIGHT)
    for _ in range(3):
        b = vs.ba_solve(*_args(w))
        assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
        assert np.array_equal(a["chi2_trace"], b["chi2_trace"])


def test_large_windows_take_the_global_memory_paths(vs, oracle):
    # 18 free cameras -> 108 x 108 (slab in HBM, factorisation in LDS); 25 / 40 free cameras -> 150 x 150 / 240 x 240
    # (both in HBM: blocked factorisation with 24-column panels, the last one partial)
    for n_cams in (19, 26, 41):
        w = ba_workload(n_cams=n_cams, n_points=300, seed=n_cams, visibility=0.5)
        _compare(vs.ba_solve(*_args(w), max_iterations=5), oracle.ba_solve(*_args(w), max_iterations=5))


def test_noise_free_converges_to_ground_truth(vs):
    w = ba_workload(n_cams=5, n_points=60, seed=7, noise_px=0, outlier_frac=0)
    w["pose_fixed"][1] = 1
    w["poses"][1] = w["poses_gt"][1]
    g = vs.ba_solve(*_args(w), max_iterations=40)
    assert g["chi2_final"] < 1e-12 * g["chi2_initial"]
    assert np.allclose(g["poses"], w["poses_gt"], atol=1e-6) and np.allclose(g["points"], w["points_gt"], atol=1e-5)


def _oracle_sensitivity(oracle, w, max_iterations, reps=3):
    """How far the ORACLE's own LM path moves under perturbations that leave the problem the same to rounding: per-trial
    and per-iteration relative spread of chi2 over runs with (a) the observations, (b) the points, (c) the poses moved by
    one ulp (+-1.1e-16 relative) and (d) the observations summed in another order (a permutation of the edge list).  This is
    the conditioning of the scene's LM path -- an implementation with a different (equally valid) summation order cannot
    be expected to stay closer to the oracle than the oracle stays to itself.  Round 2 perturbed the observations only: that
    does not touch the state the normal equations are built at and under-reports the spread by three orders of magnitude
    on some scenes (seed 6, trial 0: 2.4e-14 against 4.8e-11 for a mere reordering of the edges -- and 4.8e-11 is exactly
    how far the HIP solver is from the oracle there; profiles/r03_seed6_sensitivity.txt)."""
    base = oracle.ba_solve(*_args(w), max_iterations=max_iterations)
    rng = np.random.default_rng(1234)
    it_spread = np.zeros(len(base["chi2_trace"]))
    tr_spread = np.zeros(len(base["trial_trace"]))
    same_path = True

    def ulp(x):
        return x * (1.0 + (rng.integers(0, 2, x.shape) * 2 - 1) * 1.1e-16)

    def variants():
        for _ in range(reps):
            yield dict(w, obs_uv=ulp(w["obs_uv"]))
            yield dict(w, points=ulp(w["points"]))
            yield dict(w, poses=ulp(w["poses"]))
            p = rng.permutation(len(w["obs_pose"]))
            yield dict(w, obs_pose=w["obs_pose"][p], obs_point=w["obs_point"][p], obs_uv=w["obs_uv"][p])
    for w2 in variants():
        o2 = oracle.ba_solve(*_args(w2), max_iterations=max_iterations)
        n = min(len(it_spread), len(o2["chi2_trace"]))
        it_spread[:n] = np.maximum(it_spread[:n], np.abs(o2["chi2_trace"][:n] - base["chi2_trace"][:n]) / np.abs(base["chi2_trace"][:n]))
        m = min(len(tr_spread), len(o2["trial_trace"]))
        a, b = o2["trial_trace"][:m, 1], base["trial_trace"][:m, 1]
        fin = np.isfinite(a) & np.isfinite(b) & (np.abs(b) < 1e300)
        tr_spread[:m][fin] = np.maximum(tr_spread[:m][fin], np.abs(a[fin] - b[fin]) / np.abs(b[fin]))
        same_path &= o2["trials"] == base["trials"]
    return base, it_spread, tr_spread, same_path


@pytest.mark.parametrize("seed,st,sd,sp", [(3, 1.5, 40, 1.5), (6, 1.5, 40, 1.5), (7, 0.8, 25, 1.0), (1, 0.8, 25, 1.0)])
def test_rejected_steps_and_termination(vs, oracle, seed, st, sd, sp):
    """A terrible start forces rejected trials (lambda growth, restore) in both implementations; seed 1 also terminates.
    Lock step is asserted PER TRIAL (lambda used, trial chi2, gain ratio, Cholesky verdict) -- not only per iteration --
    with a tolerance derived from the scene's own conditioning: the oracle re-run on inputs perturbed by one ulp and on
    a reordered edge list (_oracle_sensitivity).  Round 1 found seed 7 apart by 1.4e-7 at iteration 1 while seeds 1/3/6 stay at 1e-10..1e-14;
    the oracle itself moves by 1e-7..2e-7 there under a one-ulp perturbation (profiles/r02_seed7_sensitivity.txt), i.e.
    the scene is ill-conditioned, the reject/restore path is not at fault: every accept/reject decision and every
    Cholesky verdict below is identical."""
    w = ba_workload(n_cams=4, n_points=50, seed=seed, pose_sigma_t=st, pose_sigma_deg=sd, point_sigma=sp)
    g = vs.ba_solve(*_args(w), max_iterations=15, trial_trace=True)
    o, it_spread, tr_spread, same_path = _oracle_sensitivity(oracle, w, 15)
    assert o["trials"] > o["iterations"] and g["trials"] > g["iterations"]
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], o["poses"]))
    print("wild start seed %d: worst relative pose difference %.2e, chi2 %.6f vs %.6f; oracle self-spread per iteration "
          "(1-ulp inputs) max %.1e" % (seed, rel, g["chi2_final"], o["chi2_final"], it_spread.max()))
    assert rel <= CONTRACT
    assert np.isclose(g["chi2_final"], o["chi2_final"], rtol=1e-2)
    # iteration 0 starts from identical states: tight
    assert np.isclose(g["chi2_trace"][0], o["chi2_trace"][0], rtol=1e-9)
    # per-trial lock step while the oracle's own path is reproducible (spread < 1e-3): decisions identical, values within
    # 20 x the oracle's self-spread (floor 1e-12: the spread is a maximum over a dozen runs, not a bound)
    gt, ot = g["trial_trace"], o["trial_trace"]
    n = min(len(gt), len(ot))
    live = n
    # ... and while the gain ratio means something: once LM has converged rho is rounding noise (+-1e-9) whose sign is
    # not reproducible, and with it the accept/reject decision
    bad = np.nonzero((tr_spread[:n] > 1e-3) | (np.abs(ot[:n, 2]) < 1e-6))[0]
    if len(bad):
        live = int(bad[0])
    assert live >= 3
    assert np.array_equal(gt[:live, 3], ot[:live, 3])                       # Cholesky verdict of every trial
    assert np.array_equal(gt[:live, 2] > 0, ot[:live, 2] > 0)               # accept / reject decision of every trial
    for k in range(live):
        tol = max(1e-12, 20.0 * tr_spread[:k + 1].max())
        assert np.isclose(gt[k, 0], ot[k, 0], rtol=tol), ("lambda", k, gt[k], ot[k], tol)
        if np.isfinite(ot[k, 1]) and abs(ot[k, 1]) < 1e300:
            assert np.isclose(gt[k, 1], ot[k, 1], rtol=tol), ("trial chi2", k, gt[k], ot[k], tol)
    if same_path and live == n:
        assert g["trials"] == o["trials"] and g["iterations"] == o["iterations"] and g["terminated"] == o["terminated"]
    # per-iteration traces within the same band
    for k in range(min(len(g["chi2_trace"]), len(o["chi2_trace"]), 4)):
        tol = max(1e-12, 20.0 * it_spread[:k + 1].max())
        assert np.isclose(g["chi2_trace"][k], o["chi2_trace"][k], rtol=tol), (k, g["chi2_trace"][k], o["chi2_trace"][k], tol)


def test_not_positive_definite_trials_in_lock_step(vs, oracle):
    """The reference's one real edge case (debug.txt: g2o dumped an indefinite reduced camera system and rejected the
    trial).  Negative-definite information on a third of the edges makes H indefinite at lambda_0, so the first
    trials fail in the Cholesky, lambda grows (x2, x4, ...) until the damped system is positive definite: the GPU must
    fail and recover at exactly the same trials as the oracle."""
    w = ba_workload(n_cams=4, n_points=60, seed=41)
    n = len(w["obs_pose"])
    rng = np.random.default_rng(41)
    info = np.tile([1.0, 0.0, 1.0], (n, 1))
    info[rng.random(n) < 0.3] = [-1.0, 0.0, -1.0]
    g = vs.ba_solve(*_args(w), obs_info=info, max_iterations=2, trial_trace=True)
    o = oracle.ba_solve(*_args(w), obs_info=info, max_iterations=2)
    assert o["not_pd"] >= 3 and o["trials"] > o["not_pd"]          # the scene does what it is meant to do
    assert g["not_pd"] == o["not_pd"] and g["trials"] == o["trials"] and g["iterations"] == o["iterations"]
    assert np.array_equal(g["trial_trace"][:, 3], o["trial_trace"][:, 3])
    assert np.allclose(g["trial_trace"][:, 0], o["trial_trace"][:, 0], rtol=1e-9)      # lambda schedule
    first_ok = int(np.argmax(o["trial_trace"][:, 3] > 0))
    assert first_ok >= 3 and np.all(g["trial_trace"][:first_ok, 1] > 1e300)           # failed trials carry DBL_MAX
    assert np.isclose(g["trial_trace"][first_ok, 1], o["trial_trace"][first_ok, 1], rtol=1e-7)
    # a scene that never becomes positive definite: 10 failed trials, terminated, state untouched
    info[:] = [-1.0, 0.0, -1.0]
    g = vs.ba_solve(*_args(w), obs_info=info, huber_delta=0, max_iterations=3)
    o = oracle.ba_solve(*_args(w), obs_info=info, huber_delta=0, max_iterations=3)
    assert (g["not_pd"], g["trials"], g["terminated"], g["iterations"]) == (o["not_pd"], o["trials"], o["terminated"], o["iterations"])
    if o["not_pd"] == o["trials"]:
        assert np.allclose(g["poses"], w["poses"], atol=1e-12) and np.array_equal(g["points"], w["points"])


def test_device_cholesky_rejects_the_reference_debug_matrix(vs, oracle):
    """tests/golden/reference_debug_matrix.txt = the reference's debug.txt: the 90 x 90 reduced camera system (15 poses x
    6) g2o dumped on a Cholesky failure -- symmetric, one negative eigenvalue.  The DEVICE factorisation (the kernels
    vs_ba_solve uses, reached through the vs_ba_debug_cholesky hook) must reject it, accept it once shifted, and solve
    the shifted system like LAPACK; both for the one-workgroup LDS path (n = 90) and the blocked HBM path (n = 180)."""
    import os
    from conftest import GOLDEN
    rows = [ln.split() for ln in open(os.path.join(GOLDEN, "reference_debug_matrix.txt")) if ln[0] not in "#\n"]
    A = np.zeros((90, 90))
    for r_, c_, v in rows:
        A[int(r_) - 1, int(c_) - 1] = float(v)
    ev = np.linalg.eigvalsh(A)
    assert ev[0] < 0 < ev[1]
    rng = np.random.default_rng(3)
    b = rng.standard_normal(90) * 1e10
    ok, _ = vs.debug_cholesky(A, b)
    assert not ok and oracle.cholesky_lower(A)[1] != 0
    B = A + (1e-3 * ev[-1] - ev[0]) * np.eye(90)
    ok, x = vs.debug_cholesky(B, b)
    assert ok and oracle.cholesky_lower(B)[1] == 0
    ref = np.linalg.solve(B, b)
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    # the reference's window sizes (n <= 60): leading blocks, definite and with a negative pivot late in the factorisation
    for n in (6, 30, 54, 60):
        Bn = B[:n, :n]
        ok, x = vs.debug_cholesky(Bn, b[:n])
        refn = np.linalg.solve(Bn, b[:n])
        assert ok and np.linalg.norm(x - refn) <= 1e-9 * np.linalg.norm(refn), n
        Cn = Bn.copy()
        Cn[n - 2, n - 2] = -abs(Cn[n - 2, n - 2])      # a negative pivot late in the factorisation
        ok, _ = vs.debug_cholesky(Cn, b[:n])
        assert not ok and oracle.cholesky_lower(Cn)[1] != 0, n
    # blocked path: block-diagonal 180 x 180 with the indefinite block last (the failing pivot sits in the 4th panel)
    C2 = np.zeros((180, 180))
    C2[:90, :90] = B
    C2[90:, 90:] = A
    ok, _ = vs.debug_cholesky(C2, np.concatenate([b, b]))
    assert not ok
    C2[90:, 90:] = B * 0.5
    ok, x = vs.debug_cholesky(C2, np.concatenate([b, b]))
    assert ok
    ref2 = np.linalg.solve(C2, np.concatenate([b, b]))
    assert np.linalg.norm(x - ref2) <= 1e-9 * np.linalg.norm(ref2)


def test_everything_fixed_and_zero_iterations(vs, oracle):
    w = ba_workload(n_cams=3, n_points=10, seed=2)
    g = vs.ba_solve(*_args(w), max_iterations=0)
    o = oracle.ba_solve(*_args(w), max_iterations=0)
    assert g["iterations"] == 0 and np.isclose(g["chi2_initial"], o["chi2_initial"], rtol=1e-12)
    assert np.allclose(g["poses"], w["poses"], atol=1e-12)
    w["pose_fixed"][:] = 1
    w["point_fixed"][:] = 1
    g = vs.ba_solve(*_args(w))
    assert g["iterations"] == 0 and np.allclose(g["poses"], w["poses"], atol=1e-12)


def test_deterministic(vs):
    w = ba_workload(n_cams=6, n_points=400, seed=77, visibility=0.7)
    a = vs.ba_solve(*_args(w))
    b = vs.ba_solve(*_args(w))
    assert np.array_equal(a["poses"], b["poses"]) and np.array_equal(a["points"], b["points"])
    assert np.array_equal(a["chi2_trace"], b["chi2_trace"])


def test_single_tile_schur_kernel_equals_the_tile_kernel(vs):
    """Windows of <= 10 free cameras take ba_schur_small; it keeps ba_schur_tile's accumulator ownership, product
    expressions and point order, so with the same partition of the points into slabs the whole LM run is bit-identical; other partitions and multi-batch workgroups change only the
    order in which slab sums are added.  The same holds for the linearisation of accepted states, which this path computes
    inside the trial kernel (points) and next to the Schur workgroups (cameras) instead of in a launch of its own."""

    def solve(w):
        r = vs.ba_solve(*_args(w), max_iterations=6)
        return r["poses"], r["points"], np.array(r["chi2_trace"])

    try:
        # 1540 points: the tile kernel cuts them into 256 slabs of 7 (36 of them empty), ba_schur_small at 7 points per
        # workgroup into 220 -- the same partition, also under the one-workgroup-per-CU cap of the folded path
        w = ba_workload(n_cams=10, n_points=1540, visibility=1.0)
        vs.tune_ba(1, 0, 0)
        ref = solve(w)
        for variant in (2, 0):  # 2: ba_schur_small + a linearisation launch per iteration; 0: linearisation folded into the trial
            vs.tune_ba(variant, 7, 512)
            got = solve(w)
            assert all(np.array_equal(a, b) for a, b in zip(ref, got)), variant
        for (nc, npts, vis, per, cap) in [(6, 333, 0.6, 8, 512), (10, 1500, 0.8, 8, 16), (3, 17, 1.0, 8, 512), (10, 2000, 0.9, 5, 64), (10, 2000, 1.0, 8, 512)]:
            w = ba_workload(n_cams=nc, n_points=npts, visibility=vis, seed=nc + npts)
            vs.tune_ba(1, 0, 0)
            ref = solve(w)
            vs.tune_ba(0, per, cap)
            got = solve(w)
            for a, b in zip(ref, got):
                assert np.allclose(a, b, rtol=1e-9, atol=1e-10), (nc, npts, vis, per, cap)
    finally:
        vs.tune_ba(0, 8, 512)


def test_one_launch_motion_only_solve_equals_the_launch_per_step_form(vs):
    """Motion-only windows of up to 64 cameras run the whole LM solve in ONE launch (ba_motion_persistent: observations in
    registers -- beyond 1024 per camera re-read from memory -- and a mailbox rendezvous after every step); operands, order
    of operations and the decision are those of the launch-per-step kernel, so the results are bit-identical -- and
    windows of more cameras keep taking the launch-per-step form."""

    def solve(w, **kw):
        fixed = np.ones(len(w["points"]), np.uint8)
        r = vs.ba_solve(w["poses"], w["pose_fixed"], w["points"], fixed, w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"], **kw)
        return r["poses"], np.array(r["chi2_trace"]), np.array([r["iterations"], r["trials"]])

    try:
        for (nc, npts, vis, seed, kw) in [(8, 500, 0.7, 9, {}), (2, 40, 1.0, 3, {}), (19, 420, 1.0, 5, {"max_iterations": 10}),
                                          (30, 1000, 0.9, 7, {"huber_delta": 0.0}), (6, 1500, 1.0, 11, {})]:
            w = ba_workload(n_cams=nc, n_points=npts, seed=seed, point_sigma=0, visibility=vis)
            vs.tune_ba(motion_variant=1)
            ref = solve(w, **kw)
            vs.tune_ba(motion_variant=0)
            got = solve(w, **kw)
            assert all(np.array_equal(a, b) for a, b in zip(ref, got)), (nc, npts)
            assert ref[2][1] >= 1
    finally:
        vs.tune_ba(motion_variant=0)


def test_one_launch_motion_only_solve_at_its_limits(vs, oracle):
    """64 cameras is the largest window the one-launch form takes (65 falls back to a launch per step), 1024 observations
    per camera the most its threads keep in registers (1025: the instantiation that re-reads the surplus from memory);
    information matrices ride along in registers too.  All of them against the oracle, and the two forms against each
    other."""
    rng = np.random.default_rng(3)
    for (nc, npts, vis, info) in [(65, 300, 0.5, False), (66, 120, 1.0, False), (5, 1024, 1.0, False), (5, 1025, 1.0, False),
                                  (7, 200, 0.9, True)]:
        w = ba_workload(n_cams=nc, n_points=npts, seed=nc + npts, point_sigma=0, visibility=vis)
        fixed = np.ones(len(w["points"]), np.uint8)
        kw = {}
        if info:
            a = rng.uniform(0.5, 2.0, len(w["obs_pose"]))
            b = rng.uniform(-0.2, 0.2, len(w["obs_pose"]))
            kw["obs_info"] = np.stack([a, b, a + 0.3], axis=1)
        args = (w["poses"], w["pose_fixed"], w["points"], fixed, w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
        try:
            vs.tune_ba(motion_variant=1)
            ref = vs.ba_solve(*args, **kw)
            vs.tune_ba(motion_variant=0)
            got = vs.ba_solve(*args, **kw)
        finally:
            vs.tune_ba(motion_variant=0)
        assert np.array_equal(ref["poses"], got["poses"]) and ref["trials"] == got["trials"], (nc, npts)
        o = oracle.ba_solve(*args, **kw)  # at convergence the accept / reject pattern is rounding noise: compare the optimum
        assert abs(got["chi2_final"] - o["chi2_final"]) <= 1e-9 * max(1.0, o["chi2_final"]), (nc, npts)
        assert np.allclose(got["poses"], o["poses"], rtol=0, atol=1e-7), (nc, npts)


def test_observation_order_does_not_matter_beyond_rounding(vs, oracle):
    """vs_ba_solve groups the observations by point on the host (stable); callers that add them point by point (the
    reference, LocalBA.py:164-172) hit the identity-order fast path, any other order goes through the counting sort.  A
    shuffled copy of cfg4 must agree with the oracle on the same shuffled input and with the ordered run."""
    w = ba_workload()
    rng = np.random.default_rng(17)
    perm = rng.permutation(len(w["obs_pose"]))
    ws = dict(w, obs_pose=w["obs_pose"][perm], obs_point=w["obs_point"][perm], obs_uv=w["obs_uv"][perm])
    g_sorted = vs.ba_solve(*_args(w), max_iterations=5)
    g_shuf = vs.ba_solve(*_args(ws), max_iterations=5)
    o_shuf = oracle.ba_solve(*_args(ws), max_iterations=5)
    _compare(g_shuf, o_shuf)
    assert np.allclose(g_shuf["poses"], g_sorted["poses"], rtol=0, atol=1e-9)
    assert np.allclose(g_shuf["points"], g_sorted["points"], rtol=0, atol=1e-9)


def test_one_launch_motion_only_solve_is_deterministic(vs):
    """The camera workgroups of ba_motion_persistent exchange their partials through polled mailboxes -- the timing of the
    exchange varies from run to run, the arithmetic must not: twenty solves of one window, identical bits."""
    w = ba_workload(n_cams=19, n_points=430, seed=2, point_sigma=0, visibility=1.0)
    fixed = np.ones(len(w["points"]), np.uint8)
    args = (w["poses"], w["pose_fixed"], w["points"], fixed, w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    ref = vs.ba_solve(*args)
    for _ in range(20):
        got = vs.ba_solve(*args)
        assert np.array_equal(ref["poses"], got["poses"]) and np.array_equal(ref["chi2_trace"], got["chi2_trace"])
        assert ref["trials"] == got["trials"]


def test_scale_edges_with_the_camera_role_split_over_workgroups(vs, oracle):
    """More than 512 observations per camera: several workgroups share a camera in the linearisation's camera role; with
    scale edges the last of them to arrive assembles the block row (the EdgeSBAScale terms live there), without them
    ba_reduce adds the parts.  Both against the oracle, on a single-tile window (the folded path) and on a two-tile one."""
    for nc in (5, 14):
        w = ba_workload(n_cams=nc, n_points=900, seed=40 + nc, noise_px=0.3, outlier_frac=0.02)
        meas = [np.linalg.norm(w["poses_gt"][i][:3, 3] - w["poses_gt"][i - 1][:3, 3]) * (1.0 + 0.03 * i) for i in range(1, nc)]
        se = (list(range(0, nc - 1)), list(range(1, nc)), meas)
        _compare(vs.ba_solve(*_args(w), scale_edges=se), oracle.ba_solve(*_args(w), scale_edges=se))
        _compare(vs.ba_solve(*_args(w)), oracle.ba_solve(*_args(w)))


def test_large_problem_structure_passes_on_several_host_threads(vs):
    """Problems with >= 400 000 observations build their structure (counts, active points, Hpl blocks, per-camera lists) on
    several host threads -- each takes a range of the point-grouped observation list.  The same scene with its
    observations shuffled is not grouped: it falls back to the sequential passes (and the sorted copy).  Both must give
    the same solve; and the threaded one must still reject a bad index."""
    r = np.random.default_rng(11)
    n_cams, n_points, window = 40, 45000, 10
    fx, fy, cx, cy = ICL_NUIM_K
    poses = np.tile(np.eye(4), (n_cams, 1, 1))
    poses[:, 0, 3] = 0.1 * np.arange(n_cams)
    start = r.integers(0, n_cams - window + 1, n_points)
    pts = np.stack([0.1 * (start + window / 2) + r.uniform(-1, 1, n_points), r.uniform(-1.2, 1.2, n_points), r.uniform(2.5, 5.5, n_points)], 1)
    cam = (start[:, None] + np.arange(window)[None, :]).astype(np.int32).ravel()
    pt = np.repeat(np.arange(n_points, dtype=np.int32), window)
    pc = pts[pt] - poses[cam, :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1) + r.normal(0, 0.5, (len(cam), 2))
    poses0 = poses.copy()
    poses0[1:, :3, 3] += r.normal(0, 0.01, (n_cams - 1, 3))
    pts0 = pts + r.normal(0, 0.03, pts.shape)
    fixed = np.zeros(n_cams, np.uint8)
    fixed[0] = 1
    pfix = np.zeros(n_points, np.uint8)
    pfix[::97] = 1  # some fixed points: inactive where only fixed cameras see them
    assert len(cam) >= 400000
    a = vs.ba_solve(poses0, fixed, pts0, pfix, cam, pt, uv, ICL_NUIM_K, max_iterations=2)
    perm = r.permutation(len(cam))
    b = vs.ba_solve(poses0, fixed, pts0, pfix, cam[perm], pt[perm], uv[perm], ICL_NUIM_K, max_iterations=2)
    assert a["trials"] == b["trials"] and np.allclose(a["chi2_trace"], b["chi2_trace"], rtol=1e-10)
    assert np.allclose(a["poses"], b["poses"], rtol=0, atol=1e-10) and np.allclose(a["points"], b["points"], rtol=0, atol=1e-9)
    # observation arrays in pinned memory are DMA-ed from where they lie instead of being copied into the arena first
    c = vs.ba_solve(poses0, fixed, pts0, pfix, vs.pin(cam), vs.pin(pt), vs.pin(uv), ICL_NUIM_K, max_iterations=2)
    assert np.array_equal(a["poses"], c["poses"]) and np.array_equal(a["points"], c["points"]) and np.array_equal(a["chi2_trace"], c["chi2_trace"])
    bad = cam.copy()
    bad[len(bad) // 2] = n_cams
    with pytest.raises(Exception):
        vs.ba_solve(poses0, fixed, pts0, pfix, bad, pt, uv, ICL_NUIM_K, max_iterations=1)


def _sliding_window_scene(n_cams, n_points, window, seed):
    """cameras on a 0.1 m-spaced track, every point seen from `window` consecutive cameras starting at a random one: the
    co-visibility of key-frame bundle adjustment (a banded reduced system)."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy = ICL_NUIM_K
    poses = np.tile(np.eye(4), (n_cams, 1, 1))
    poses[:, 0, 3] = 0.1 * np.arange(n_cams)
    start = r.integers(0, n_cams - window + 1, n_points)
    pts = np.stack([0.1 * (start + window / 2) + r.uniform(-1, 1, n_points), r.uniform(-1.2, 1.2, n_points), r.uniform(2.5, 5.5, n_points)], 1)
    cam = (start[:, None] + np.arange(window)[None, :]).astype(np.int32).ravel()
    pt = np.repeat(np.arange(n_points, dtype=np.int32), window)
    pc = pts[pt] - poses[cam, :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1) + r.normal(0, 0.5, (len(cam), 2))
    bad = r.random(len(uv)) < 0.02
    uv[bad] += r.uniform(-50, 50, (int(bad.sum()), 2))
    poses0 = poses.copy()
    poses0[1:, :3, 3] += r.normal(0, 0.01, (n_cams - 1, 3))
    fixed = np.zeros(n_cams, np.uint8)
    fixed[0] = 1
    return dict(poses=poses0, pose_fixed=fixed, points=pts + r.normal(0, 0.03, pts.shape), point_fixed=np.zeros(n_points, np.uint8),
                obs_pose=cam, obs_point=pt, obs_uv=uv, K=ICL_NUIM_K, poses_gt=poses)


def _solve_both_ways(vs, w, iters=2):
    """the same large problem with its structure built on the device and by the host passes; (device result, host result)"""
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], vs.pin(w["obs_pose"]), vs.pin(w["obs_point"]), vs.pin(w["obs_uv"]), w["K"])
    try:
        vs.tune_ba_structure(on_host=False)
        d = vs.ba_solve(*args, max_iterations=iters)
        on_device = vs.ba_structure_on_device()
        vs.tune_ba_structure(on_host=True)
        h = vs.ba_solve(*args, max_iterations=iters)
        assert not vs.ba_structure_on_device()
    finally:
        vs.tune_ba_structure(on_host=False)
    return d, h, on_device


def _same_solve(a, b):
    return (a["trials"] == b["trials"] and np.array_equal(a["chi2_trace"], b["chi2_trace"]) and np.array_equal(a["poses"], b["poses"])
            and np.array_equal(a["points"], b["points"]))


def _ragged_scene(seed=31, n_cams=300, n_points=70000):
    """300 cameras on a track, three of them fixed (one in the middle of the window), every point seen from 2 .. 12 cameras out of a
    neighbourhood of 14, in random camera order: ragged observation ranges, Hpl-less observations of free points everywhere."""
    r = np.random.default_rng(seed)
    fx, fy, cx, cy = ICL_NUIM_K
    poses = np.tile(np.eye(4), (n_cams, 1, 1))
    poses[:, 0, 3] = 0.1 * np.arange(n_cams)
    start = r.integers(0, n_cams - 14, n_points)
    cnt = r.integers(2, 13, n_points)
    pts = np.stack([0.1 * (start + 7) + r.uniform(-1, 1, n_points), r.uniform(-1.2, 1.2, n_points), r.uniform(2.5, 5.5, n_points)], 1)
    cam = np.concatenate([start[j] + r.permutation(14)[:cnt[j]] for j in range(n_points)]).astype(np.int32)
    pt = np.repeat(np.arange(n_points, dtype=np.int32), cnt)
    pc = pts[pt] - poses[cam, :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1) + r.normal(0, 0.5, (len(cam), 2))
    poses0 = poses.copy()
    poses0[:, :3, 3] += r.normal(0, 0.01, (n_cams, 3))
    fixed = np.zeros(n_cams, np.uint8)
    fixed[[0, n_cams // 2, n_cams - 1]] = 1
    poses0[fixed == 1] = poses[fixed == 1]
    return dict(poses=poses0, pose_fixed=fixed, points=pts + r.normal(0, 0.03, pts.shape), point_fixed=np.zeros(n_points, np.uint8),
                obs_pose=cam, obs_point=pt, obs_uv=uv, K=ICL_NUIM_K, poses_gt=poses)


@pytest.mark.parametrize("case", ["banded", "fixed_points", "unobserved_points", "dense", "single_tile", "one_point_per_camera_run", "ragged"])
def test_large_problem_structure_built_on_the_device_equals_the_host_passes(vs, case):
    """Round 4: problems with >= 400 000 observations in pinned arrays build their sparsity structure (observation ranges, Hpl
    blocks, per-camera lists, tile masks, the banded-window plan) on the device, from the device copy of the observation list
    (csrc/vs_ba_build.hip).  Integer work only, so the arrays -- and with them the whole solve -- must be BIT-identical to what
    the host passes give, over every Schur path the structure feeds."""
    n_cams, n_points, window = 40, 45000, 10
    if case == "dense":
        n_cams, n_points, window = 30, 16000, 30     # every camera sees every point: no banded window, the tile kernel
    elif case == "single_tile":
        n_cams, n_points, window = 10, 45000, 10     # nine free cameras: ba_schur_small
    w = _ragged_scene() if case == "ragged" else _sliding_window_scene(n_cams, n_points, window, seed=21)
    if case == "fixed_points":      # fixed points seen from free cameras only: active observations without an Hpl block
        seen_by_0 = np.zeros(n_points, bool)
        seen_by_0[w["obs_point"][w["obs_pose"] == 0]] = True
        w["point_fixed"][(np.arange(n_points) % 7 == 3) & ~seen_by_0] = 1
    elif case == "unobserved_points":  # free points nobody observes (damping only), in the middle and at both ends of the list
        drop = (w["obs_point"] % 1000 == 5) | (w["obs_point"] < 3) | (w["obs_point"] >= n_points - 4)
        for k in ("obs_pose", "obs_point", "obs_uv"):
            w[k] = np.ascontiguousarray(w[k][~drop])
    elif case == "one_point_per_camera_run":  # observations of a point not in camera order
        r = np.random.default_rng(5)
        o = np.arange(len(w["obs_pose"])).reshape(n_points, window)
        o = np.take_along_axis(o, r.permuted(np.tile(np.arange(window), (n_points, 1)), axis=1), 1).ravel()
        for k in ("obs_pose", "obs_point", "obs_uv"):
            w[k] = np.ascontiguousarray(w[k][o])
    assert len(w["obs_pose"]) >= 400000
    d, h, on_device = _solve_both_ways(vs, w)
    assert on_device, "the device-side structure was not taken"
    assert d["trials"] >= 2 and d["chi2_final"] < d["chi2_initial"]
    assert _same_solve(d, h)


def test_device_structure_over_repeated_calls_of_changing_size(vs):
    """the arena and the flag words are reused from call to call: scenes of different sizes, a fall-back in between, the first scene
    again -- every device-built solve equals the host-built one, and the first result comes back bit for bit"""
    scenes = [_sliding_window_scene(40, 45000, 10, seed=41), _ragged_scene(seed=42, n_cams=120, n_points=60000), _sliding_window_scene(25, 50000, 9, seed=43)]
    first = None
    for k in (0, 1, 2, 0):
        d, h, on_device = _solve_both_ways(vs, scenes[k])
        assert on_device and _same_solve(d, h), k
        if k == 0 and first is None:
            first = d
            bad = dict(scenes[0], obs_point=scenes[0]["obs_point"][::-1].copy(), obs_pose=scenes[0]["obs_pose"][::-1].copy(), obs_uv=scenes[0]["obs_uv"][::-1].copy())
            d2, h2, dev2 = _solve_both_ways(vs, bad)   # descending points: not grouped in ascending order -> host passes
            assert not dev2 and _same_solve(d2, h2)
    assert _same_solve(d, first)


def test_large_problems_the_device_structure_does_not_cover_fall_back_to_the_host_passes(vs):
    """... a list that is not grouped by point, inactive observations (fixed point seen from the fixed camera), the same camera
    twice in one point: the device reports it in one word and the host passes run after all; an index out of range is still an
    error."""
    w = _sliding_window_scene(40, 45000, 10, seed=22)
    r = np.random.default_rng(8)
    perm = r.permutation(len(w["obs_pose"]))
    shuffled = dict(w, obs_pose=w["obs_pose"][perm], obs_point=w["obs_point"][perm], obs_uv=w["obs_uv"][perm])
    d, h, on_device = _solve_both_ways(vs, shuffled)
    assert not on_device and _same_solve(d, h)
    inactive = dict(w, point_fixed=w["point_fixed"].copy())
    inactive["point_fixed"][::97] = 1
    d, h, on_device = _solve_both_ways(vs, inactive)
    assert not on_device and _same_solve(d, h)
    twice = dict(w, obs_pose=w["obs_pose"].copy())
    twice["obs_pose"][10 * 20000 + 3] = twice["obs_pose"][10 * 20000 + 2]   # point 20000 seen twice from one camera
    d, h, on_device = _solve_both_ways(vs, twice)
    assert not on_device and _same_solve(d, h)
    for where in (0, len(w["obs_pose"]) // 2, len(w["obs_pose"]) - 1):
        bad = dict(w, obs_point=w["obs_point"].copy())
        bad["obs_point"][where] = 45000
        with pytest.raises(Exception):
            _solve_both_ways(vs, bad, iters=1)
    bad = dict(w, obs_pose=w["obs_pose"].copy())
    bad["obs_pose"][12345] = -1
    with pytest.raises(Exception):
        _solve_both_ways(vs, bad, iters=1)


def test_banded_windows_on_the_matrix_cores_agree_with_the_tile_kernel_and_the_oracle(vs, oracle):
    """Windows of more than ten free cameras whose points are seen from neighbouring cameras only: the points are ordered
    by their lowest camera, every slab's contribution is one dense window of S accumulated with FP64 MFMA
    (ba_schur_window), the slabs are summed by ba_reduce_window and -- without scale edges -- the banded system is
    factorised by ba_chol_band in one launch.  vs_tune_ba variant 3 keeps such problems on the tile kernel and the dense
    factorisation: both must agree with the oracle, and with each other to rounding."""
    cases = [(24, 3000, 10, 1), (100, 1500, 10, 6), (40, 4000, 14, 2), (13, 400, 5, 5), (12, 30, 4, 8), (17, 70, 16, 9)]
    try:
        for n_cams, n_points, window, seed in cases:
            w = _sliding_window_scene(n_cams, n_points, window, seed)
            w["point_fixed"][::53] = 1                    # some fixed points
            w["obs_pose"][w["obs_point"] == 7] = 0        # a free point seen from the fixed camera only (duplicates too)
            vs.tune_ba(schur_variant=0)
            a = vs.ba_solve(*_args(w), max_iterations=4)
            a2 = vs.ba_solve(*_args(w), max_iterations=4)   # fixed summation orders everywhere: run to run the same bits
            assert np.array_equal(a["poses"], a2["poses"]) and np.array_equal(a["points"], a2["points"])
            assert np.array_equal(a["chi2_trace"], a2["chi2_trace"])
            vs.tune_ba(schur_variant=3)
            b = vs.ba_solve(*_args(w), max_iterations=4)
            o = oracle.ba_solve(*_args(w), max_iterations=4)
            _compare(a, o)
            _compare(b, o)
            assert np.abs(a["poses"] - b["poses"]).max() < 1e-11 and np.abs(a["points"] - b["points"]).max() < 1e-11
        # with scale edges the camera Hessian is no longer banded: windowed Schur complement, dense factorisation
        w = _sliding_window_scene(30, 2500, 10, 9)
        idx = np.arange(1, 30)
        meas = [float(np.linalg.norm(w["poses_gt"][i][:3, 3] - w["poses_gt"][i - 1][:3, 3]) * (1.0 + 0.01 * (i % 3))) for i in idx]
        se = ((idx - 1).tolist(), idx.tolist(), meas)
        se_far = ([0, 2], [29, 25], [2.95, 2.31])     # ... including pairs far outside any window
        for edges in (se, se_far):
            vs.tune_ba(schur_variant=0)
            _compare(vs.ba_solve(*_args(w), max_iterations=4, scale_edges=edges), oracle.ba_solve(*_args(w), max_iterations=4, scale_edges=edges))
    finally:
        vs.tune_ba(schur_variant=0)


def test_banded_cholesky_is_bitwise_the_dense_factorisation(vs):
    """ba_chol_band (one launch, a window of the band sliding through LDS) performs the dense panel / update kernels'
    subtractions in their order and skips only exact zeros: same bits, also for a band that is not a multiple of the panel
    width and one wider than the kernel takes (dense path both times); an indefinite matrix is rejected."""
    r = np.random.default_rng(0)
    try:
        # (1194 unknowns: the dense path's panels are 12 columns wide there, the banded one's always 24 -- same bits)
        for n, cams_band in ((594, 11), (594, 16), (300, 5), (132, 3), (600, 1), (594, 17), (1194, 9)):
            M = np.zeros((n, n))
            for a in range(n // 6):
                for b in range(max(0, a - cams_band + 1), a + 1):
                    M[6 * a:6 * a + 6, 6 * b:6 * b + 6] = r.normal(size=(6, 6))
            S = M + M.T
            S += np.eye(n) * (np.abs(S).sum(1).max() + 1.0)
            rhs = r.normal(size=n)
            vs.tune_ba(schur_variant=0)
            ok1, x1 = vs.debug_cholesky(S, rhs)
            vs.tune_ba(schur_variant=3)
            ok2, x2 = vs.debug_cholesky(S, rhs)
            assert ok1 and ok2 and np.array_equal(x1, x2)
            assert np.abs(x1 - np.linalg.solve(S, rhs)).max() < 1e-12
        vs.tune_ba(schur_variant=0)
        S[300, 300] = -1.0
        assert not vs.debug_cholesky(S, rhs)[0]
    finally:
        vs.tune_ba(schur_variant=0)


@pytest.mark.gpu
def test_packed_lds_solver_is_bitwise_the_other_dense_paths(vs, oracle):
    """ba_solve_block with the lower triangle packed (127 .. 198 unknowns in 160 KB of LDS, vs_tune_ba_solve): the same subtractions
    in the same order as the square layout (<= 126 unknowns) and as the blocked factorisation in HBM it replaces (ba_chol_panel /
    ba_chol_update / ba_chol_finish) -- the same bits for the solution of the reduced system; an indefinite matrix is rejected; whole
    solves agree with the oracle on either side of both size limits, and with each other to rounding."""
    r = np.random.default_rng(3)
    try:
        for n in (6, 24, 54, 126, 132, 174, 198):
            M = r.normal(size=(n, n))
            S = M @ M.T + n * np.eye(n)
            rhs = r.normal(size=n)
            got = {}
            for mode in (0, 1, 2):
                vs.tune_ba_solve(mode)
                ok, x = vs.debug_cholesky(S, rhs)
                assert ok
                got[mode] = x
            assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2]), n
            assert np.abs(got[0] - np.linalg.solve(S, rhs)).max() < 1e-10
        vs.tune_ba_solve(0)
        S[150, 150] = -1.0
        assert not vs.debug_cholesky(S, rhs)[0]
        # whole solves: 22 / 33 / 34 cameras (126 / 192 / 198 unknowns; 204 is past the packed limit)
        for n_cams, want in ((22, "ba_solve_block"), (33, "ba_solve_block"), (34, "ba_solve_block"), (35, "ba_chol_panel")):
            w = ba_workload(n_cams=n_cams, n_points=400, visibility=0.4, seed=n_cams)
            a = vs.ba_solve(*_args(w), max_iterations=4)
            path = vs.ba_last_path()
            assert path["unknowns"] == 6 * (n_cams - 1) and path["dense"] == want, path
            _compare(a, oracle.ba_solve(*_args(w), max_iterations=4))
            vs.tune_ba_solve(1)
            b = vs.ba_solve(*_args(w), max_iterations=4)
            vs.tune_ba_solve(0)
            # (the solutions of the reduced system are the same bits, above; the camera update behind them is not the same code in
            # ba_solve_block and ba_chol_finish -- reciprocal square root + Newton steps against sqrt and a division for the unit
            # quaternion -- so whole solves agree to rounding, with the same LM decisions)
            assert np.abs(a["poses"] - b["poses"]).max() < 1e-11 and np.abs(a["points"] - b["points"]).max() < 1e-11
            assert np.allclose(a["chi2_trace"], b["chi2_trace"], rtol=1e-12) and a["trials"] == b["trials"]
    finally:
        vs.tune_ba_solve(0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_cams,n_points", [(13, 20000), (16, 30000)])
def test_window_plan_fits_the_arena_of_a_fresh_context(oracle, n_cams, n_points):
    """Round-3 advisor: the arena reservation covered the tile path's slabs (np * np + np doubles each) but not the banded-window
    path's (9 312 doubles each whatever np is): with 11-20 free cameras np is small, a window plan of hundreds of slabs outgrew
    it, and the solve ended in VS_ENOMEM 'internal arena sizing error' -- unless an earlier, larger solve on the same context
    had grown the arena.  A FRESH context, few cameras, many points seen three times each: must solve and agree with the oracle."""
    from visual_slam_amd import Context, _capi
    if _capi.device_count() == 0:
        pytest.skip("no GPU in this machine")
    w = _sliding_window_scene(n_cams, n_points, 3, 21)
    ctx = Context(0)  # nothing has grown this context's arena
    try:
        g = ctx.ba_solve(*_args(w), max_iterations=2)
    finally:
        ctx.close()
    o = oracle.ba_solve(*_args(w), max_iterations=2)
    _compare(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("n_cams,n_points,window", [(14, 140000, 3), (60, 50000, 9), (10, 42000, 10)])
def test_device_built_structure_fits_the_arena_of_a_fresh_context(n_cams, n_points, window):
    """the device-side structure carves its arrays and its temporaries (keys, block histograms, window order) out of the same
    arena: on a FRESH context, first solve, with few cameras and many points (the case the round-3 reservation missed), a banded
    and a single-tile window -- and the result equals the host passes' on another fresh context"""
    from visual_slam_amd import Context, _capi
    if _capi.device_count() == 0:
        pytest.skip("no GPU in this machine")
    w = _sliding_window_scene(n_cams, n_points, window, 77)
    assert len(w["obs_pose"]) >= 400000
    out = []
    for on_host in (False, True):
        ctx = Context(0)
        try:
            ctx.tune_ba_structure(on_host=on_host)
            args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], ctx.pin(w["obs_pose"]), ctx.pin(w["obs_point"]), ctx.pin(w["obs_uv"]), w["K"])
            out.append(ctx.ba_solve(*args, max_iterations=2))
            assert ctx.ba_structure_on_device() == (not on_host)
        finally:
            ctx.close()
    assert _same_solve(out[0], out[1])


def test_large_windows_with_two_linearisations_equal_one_linearisation_per_step(vs):
    """Round 4: large banded windows keep two sets of linearisation blocks like cfg4 -- ba_point_trial linearises the trial state's
    points, an accepted step launches only the camera role, a failed or rejected one drops the blocks.  vs_tune_ba(schur_variant=2)
    takes the same kernels with a linearisation launch per step: same bits -- on a plain scene, and on one whose first trials fail
    (negative-definite information on a third of the edges makes the reduced system indefinite until lambda has grown: the
    reference's one real edge case, see test_not_positive_definite_trials_in_lock_step)."""
    w = _sliding_window_scene(40, 45000, 10, seed=33)
    n = len(w["obs_pose"])
    info = np.tile([1.0, 0.0, 1.0], (n, 1))
    info[np.random.default_rng(3).random(n) < 0.3] = [-1.0, 0.0, -1.0]
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    for kw in (dict(max_iterations=4), dict(max_iterations=3, obs_info=info)):
        try:
            vs.tune_ba(schur_variant=0)
            a = vs.ba_solve(*args, **kw)
            vs.tune_ba(schur_variant=2)
            b = vs.ba_solve(*args, **kw)
        finally:
            vs.tune_ba(schur_variant=0)
        if "obs_info" in kw:
            assert a["not_pd"] >= 2 and a["trials"] > a["not_pd"], "the scene was meant to fail its first trials and then recover"
        assert a["not_pd"] == b["not_pd"] and _same_solve(a, b)
