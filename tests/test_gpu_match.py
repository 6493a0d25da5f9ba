"""GPU parity: HIP Hamming 2-NN + ratio compaction vs the CPU oracle, through the C ABI.  Bit-exact."""
import numpy as np
import pytest

from visual_slam_amd.workloads import match_workload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,nt", [(1, 2), (3, 5), (64, 64), (255, 33), (256, 1000), (257, 129), (1000, 3000),
                                   (2500, 2999), (4097, 70)])
def test_knn2_matches_oracle(vs, oracle, nq, nt):
    q, t = match_workload(nq, nt, n_dup=min(16, nt // 4), seed=nq + nt)
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t)
    assert np.array_equal(dist, odist)
    assert np.array_equal(idx, oidx)


def test_massive_ties_low_entropy(vs, oracle):
    rng = np.random.default_rng(5)
    t = np.zeros((3000, 32), np.uint8)
    t[:, 7] = rng.integers(0, 4, 3000)
    q = np.zeros((700, 32), np.uint8)
    q[:, 7] = rng.integers(0, 4, 700)
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    # all-identical train set: answer is (0, 1) at equal distance for every query
    t[:] = 0xA5
    idx, dist = vs.hamming_knn2(q, t)
    assert np.all(idx == [0, 1]) and np.all(dist[:, 0] == dist[:, 1])


def test_extreme_distances(vs, oracle):
    q = np.zeros((2, 32), np.uint8)
    q[1] = 255
    t = np.zeros((5, 32), np.uint8)
    t[4] = 255
    t[2, 31] = 0x80
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    assert dist[1].tolist() == [0, 255] and dist[0].tolist() == [0, 0]


def test_cfg3_10k_x_10k_bit_exact(vs, oracle):
    q, t = match_workload(10000, 10000)
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t, threads=0)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    # size-independent properties: ascending pairs, distinct indices, recomputed distances agree
    assert np.all(dist[:, 0] <= dist[:, 1]) and np.all(idx[:, 0] != idx[:, 1])
    x = np.bitwise_xor(q, t[idx[:, 0]])
    assert np.array_equal(np.unpackbits(x, axis=1).sum(1), dist[:, 0])
    tie = dist[:, 0] == dist[:, 1]
    assert np.all(idx[tie, 0] < idx[tie, 1])


@pytest.mark.parametrize("ratio", [0.8, 0.5, 1.0, 0.0, 0.95])
def test_ratio_compaction_matches_oracle(vs, oracle, ratio):
    q, t = match_workload(3000, 2500, seed=9)
    mq, mt, md = vs.match_ratio(q, t, ratio)
    oq, ot, od = oracle.match_ratio(q, t, ratio)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)


def test_edge_cases(vs):
    from visual_slam_amd import VsError
    t = np.zeros((3, 32), np.uint8)
    idx, dist = vs.hamming_knn2(np.zeros((0, 32), np.uint8), t)
    assert idx.shape == (0, 2)
    mq, _, _ = vs.match_ratio(np.zeros((0, 32), np.uint8), t)
    assert mq.shape == (0,)
    with pytest.raises(VsError):  # T < 2: the reference cannot unpack (m, n) either (frame.py:30)
        vs.hamming_knn2(np.zeros((1, 32), np.uint8), t[:1])


def test_deterministic(vs):
    q, t = match_workload(5000, 4000, seed=3)
    a = vs.hamming_knn2(q, t)
    b = vs.hamming_knn2(q, t)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
