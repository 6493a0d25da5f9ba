"""The committed golden vectors (tests/golden/oracle_golden.npz, made by tests/golden/make_golden.py): the oracle must
still reproduce them (CPU), and the HIP kernels must reproduce them too (GPU).  Integer work bit-exact, BA to 1e-9."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, icl_frame
from visual_slam_amd.workloads import ba_workload

G = np.load(os.path.join(GOLDEN, "oracle_golden.npz"))
HUBER = float(np.sqrt(5.991))


def _check(impl):
    xy, sc, desc = impl.detect_describe_bgr(G["tile_bgr"], 20, 3000)
    assert np.array_equal(xy, G["tile_xy"]) and np.array_equal(sc, G["tile_score"]) and np.array_equal(desc, G["tile_desc"])
    xy, sc, desc = impl.detect_describe_bgr(icl_frame(0), 20, 3000)
    assert np.array_equal(xy, G["icl0_xy"]) and np.array_equal(sc, G["icl0_score"]) and np.array_equal(desc, G["icl0_desc"])
    assert np.array_equal(impl.detect_describe_bgr(icl_frame(0), 20, 100)[0], G["icl0_xy_cap100"])
    idx, dist = impl.hamming_knn2(G["ham_q"], G["ham_t"])
    assert np.array_equal(idx, G["ham_idx"]) and np.array_equal(dist, G["ham_dist"])
    mq, mt, md = impl.match_ratio(G["ham_q"], G["ham_t"], 0.8)
    assert np.array_equal(mq, G["ham_mq"]) and np.array_equal(mt, G["ham_mt"]) and np.array_equal(md, G["ham_md"])
    w = ba_workload(n_cams=3, n_points=20, seed=5)
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    r = impl.ba_solve(*args, huber_delta=HUBER, max_iterations=6)
    assert np.allclose(r["poses"], G["ba_small_poses"], rtol=0, atol=1e-9)
    assert np.allclose(r["points"], G["ba_small_points"], rtol=0, atol=1e-8)
    assert np.allclose(r["chi2_trace"], G["ba_small_chi2"], rtol=1e-9)
    w = ba_workload()
    args = (w["poses"], w["pose_fixed"], w["points"], w["point_fixed"], w["obs_pose"], w["obs_point"], w["obs_uv"], w["K"])
    r = impl.ba_solve(*args, huber_delta=HUBER, max_iterations=10)
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(r["poses"], G["ba_cfg4_poses"]))
    assert rel < 1e-9 and np.isclose(r["chi2_initial"], G["ba_cfg4_chi2_initial"][0], rtol=1e-12)
    assert np.allclose(r["chi2_trace"], G["ba_cfg4_chi2"], rtol=1e-9)
    assert np.allclose(r["lambda_trace"], G["ba_cfg4_lambda"], rtol=1e-7)
    assert np.allclose(r["points"][:16], G["ba_cfg4_points_head"], rtol=0, atol=1e-9)


def _check_rows_around_the_path(impl):
    from visual_slam_amd.workloads import ICL_NUIM_K
    r = impl.pnp_ransac(G["pnp_obj"], G["pnp_img"], ICL_NUIM_K, np.eye(4), seed=12)
    assert r["found"] and np.array_equal(r["inliers"], G["pnp_inliers"]) and np.abs(r["pose"] - G["pnp_pose"]).max() < 1e-9
    e = impl.essential_ransac(G["tv_x1"], G["tv_x2"], 3.0 / 480, seed=13)
    assert e["found"] and np.array_equal(e["mask"], G["tv_mask"]) and np.abs(e["E"] - G["tv_E"]).max() < 1e-9
    sel = G["tv_mask"] == 1
    rp = impl.recover_pose(G["tv_E"], G["tv_x1"][sel], G["tv_x2"][sel])
    assert np.array_equal(rp["mask"], G["tv_pose_mask"]) and np.abs(rp["R"] - G["tv_R"]).max() < 1e-12
    assert np.abs(rp["t"] - G["tv_t"]).max() < 1e-12 and np.abs(rp["X"] - G["tv_X"]).max() < 1e-9


def test_oracle_reproduces_the_golden_vectors(oracle):
    _check_rows_around_the_path(oracle)
    assert np.array_equal(oracle.fast9_score_map(oracle.gray_mean3(G["tile_bgr"]), 20, 3), G["tile_score_map"])
    _check(oracle)


@pytest.mark.gpu
def test_hip_reproduces_the_golden_vectors(vs):
    _check(vs)
    _check_rows_around_the_path(vs)
