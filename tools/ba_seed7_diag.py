"""dev tool: per-TRIAL comparison of the LM path of the 'wild start' scenes (tests/test_gpu_ba.py::
test_rejected_steps_and_termination): oracle vs the oracle on inputs perturbed by one ulp (the scene's own conditioning)
vs the HIP solver when a GPU is present.  Written for VERDICT r01 weak #4 (seed 7: chi2 apart by 1.4e-7 at iteration 1).
usage: python tools/ba_seed7_diag.py > profiles/r02_seed7_sensitivity.txt"""
import _env  # noqa: F401
import numpy as np

from oracle import oracle
from visual_slam_amd import _capi
from visual_slam_amd.workloads import ba_workload


def args(w):
    return (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])


vs = None
if _capi.device_count() > 0:
    from visual_slam_amd import Context
    vs = Context(0)
print("columns per trial: lambda used | trial chi2 | gain ratio rho | Cholesky ok")
for seed, st, sd, sp in [(3, 1.5, 40, 1.5), (6, 1.5, 40, 1.5), (7, 0.8, 25, 1.0), (1, 0.8, 25, 1.0)]:
    w = ba_workload(n_cams=4, n_points=50, seed=seed, pose_sigma_t=st, pose_sigma_deg=sd, point_sigma=sp)
    o = oracle.ba_solve(*args(w), max_iterations=15)
    rng = np.random.default_rng(1234)
    pert = []
    for _ in range(6):
        w2 = dict(w)
        w2["obs_uv"] = w["obs_uv"] * (1.0 + (rng.integers(0, 2, w["obs_uv"].shape) * 2 - 1) * 1.1e-16)
        pert.append(oracle.ba_solve(*args(w2), max_iterations=15))
    g = vs.ba_solve(*args(w), max_iterations=15, trial_trace=True) if vs is not None else None
    print("\n=== seed %d (pose sigma %.1f m / %d deg, point sigma %.1f m): oracle %d iterations, %d trials, not_pd %d"
          % (seed, st, sd, sp, o["iterations"], o["trials"], o["not_pd"]))
    print("trial |            oracle: lambda        chi2         rho  ok | oracle self-spread of chi2 (1-ulp inputs) |"
          + ("   HIP: lambda        chi2         rho  ok | rel diff chi2 HIP vs oracle" if g else ""))
    for k in range(min(o["trials"], 14)):
        r = o["trial_trace"][k]
        spread = 0.0
        for p in pert:
            if k < len(p["trial_trace"]) and abs(r[1]) < 1e300 and abs(p["trial_trace"][k, 1]) < 1e300:
                spread = max(spread, abs(p["trial_trace"][k, 1] - r[1]) / abs(r[1]))
        line = "%5d | %18.10e %14.6f %11.4e  %d | %10.2e" % (k, r[0], r[1], r[2], int(r[3]), spread)
        if g is not None and k < len(g["trial_trace"]):
            h = g["trial_trace"][k]
            d = abs(h[1] - r[1]) / abs(r[1]) if abs(r[1]) < 1e300 and abs(h[1]) < 1e300 else float("nan")
            line += "                              | %18.10e %14.6f %11.4e  %d | %10.2e" % (h[0], h[1], h[2], int(h[3]), d)
        print(line)
    it = np.abs(np.array([p["chi2_trace"][:4] for p in pert]) - o["chi2_trace"][:4]) / o["chi2_trace"][:4]
    print("per-iteration chi2 (first 4): oracle", o["chi2_trace"][:4])
    print("   oracle self-spread (max over 6 one-ulp perturbations):", it.max(0))
    if g is not None:
        print("   HIP vs oracle                                         :", np.abs(g["chi2_trace"][:4] - o["chi2_trace"][:4]) / o["chi2_trace"][:4])
