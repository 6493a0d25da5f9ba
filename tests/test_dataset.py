"""Dataset I/O + ATE: the reference's data/ICL_NUIM file formats, parsed from the committed heads."""
import os

import numpy as np

from visual_slam_amd import dataset

ICL = os.path.join(os.path.dirname(__file__), "golden", "icl_nuim")


def test_associations_and_groundtruth_formats(tmp_path):
    a = dataset.read_associations(os.path.join(ICL, "associations.txt.head20"))
    assert len(a) == 20 and a[0] == (0, "depth/0.png", 0, "rgb/0.png") and a[19][3] == "rgb/19.png"
    idx, poses = dataset.read_trajectory(os.path.join(ICL, "traj3.gt.freiburg.head20"))
    assert idx.tolist() == list(range(1, 21)) and poses.shape == (20, 4, 4)
    assert np.allclose(poses[0], [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -2.5], [0, 0, 0, 1]])
    for P in poses:
        assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-12) and np.linalg.det(P[:3, :3]) > 0
    # second line of the file: 2 -0.00126362 0.00400925 -2.49997 0.00213729 -0.000606835 -0.000311375 0.999997
    assert np.allclose(poses[1][:3, 3], [-0.00126362, 0.00400925, -2.49997])
    assert abs(poses[1][2, 1] - 2 * 0.00213729) < 1e-5          # small-angle: R[2,1] ~ 2 qx
    out = tmp_path / "t.txt"
    dataset.write_trajectory(out, idx, poses)
    idx2, poses2 = dataset.read_trajectory(out)
    assert (idx2 == idx).all() and np.allclose(poses2, poses, atol=1e-8)


def test_sequence_directory():
    seq = dataset.Sequence(ICL, associations="associations.txt.head20", groundtruth="traj3.gt.freiburg.head20")
    assert len(seq) == 20 and seq.rgb(3).shape == (480, 640, 3) and seq.gt[1].shape == (20, 4, 4)
    d = seq.depth(0)                                            # only depth/0.png is committed (the initialisation frame)
    assert d.shape == (480, 640) and 0.5 < np.median(d) < 10
    bare = dataset.Sequence(ICL, associations="missing.txt")   # falls back to the rgb directory in numeric order
    assert [f[2] for f in bare.frames] == list(range(20)) + [150]   # 150: the wide-baseline frame of the two-view test


def test_umeyama_recovers_a_similarity_and_ate_is_zero_for_it():
    r = np.random.default_rng(0)
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec([0.3, -0.2, 0.5]).as_matrix()
    src = r.normal(size=(50, 3))
    dst = 2.5 * (R @ src.T).T + np.array([1.0, -2.0, 0.5])
    s, Rr, t = dataset.umeyama(src, dst)
    assert abs(s - 2.5) < 1e-12 and np.allclose(Rr, R, atol=1e-12) and np.allclose(t, [1, -2, 0.5], atol=1e-12)
    est = np.tile(np.eye(4), (50, 1, 1))
    gt = est.copy()
    est[:, :3, 3], gt[:, :3, 3] = src, dst
    a = dataset.ate_rmse(est, gt)
    assert a["rmse"] < 1e-12 and abs(a["scale"] - 2.5) < 1e-12
    assert dataset.ate_rmse(est, gt, with_scale=False)["rmse"] > 0.1       # SE(3) cannot absorb the scale
    gt[7, :3, 3] += [0.3, 0, 0]
    assert 0.03 < dataset.ate_rmse(est, gt)["rmse"] < 0.06 and dataset.ate_rmse(est, gt)["max"] > 0.2
    # reflections are not rotations: a mirrored trajectory must not align perfectly
    mir = est.copy()
    mir[:, 0, 3] *= -1
    assert dataset.ate_rmse(mir, est)["rmse"] > 0.1


def test_driver_trajectory_against_ground_truth(oracle):
    """End to end on the CPU back ends: the 20-frame driver trajectory aligned to ICL-NUIM's ground truth."""
    from test_slam_driver import _run, oracle_backends
    r = _run(oracle_backends(oracle))
    _, gt = dataset.read_trajectory(os.path.join(ICL, "traj3.gt.freiburg.head20"))
    a = dataset.ate_rmse(r["poses"], gt)
    assert a["path_length"] > 0.03
    assert a["rmse"] < 0.1 * a["path_length"], a      # 20 frames, ~5 cm of motion: the error stays a small fraction of it
