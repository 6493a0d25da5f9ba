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


def test_oracle_reproduces_the_golden_vectors(oracle):
    assert np.array_equal(oracle.fast9_score_map(oracle.gray_mean3(G["tile_bgr"]), 20, 3), G["tile_score_map"])
    _check(oracle)


@pytest.mark.gpu
def test_hip_reproduces_the_golden_vectors(vs):
    _check(vs)
