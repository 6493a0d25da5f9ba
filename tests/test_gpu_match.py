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


def test_cfg5_100k_x_100k_bit_exact(vs, oracle):
    """BASELINE.json configs[4] on one GPU (the sharded variant splits exactly these queries): 1e10 distance
    evaluations, compared row for row with the multi-threaded oracle."""
    q, t = match_workload(100000, 100000)
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t, threads=0)
    assert np.array_equal(dist, odist) and np.array_equal(idx, oidx)
    tie = dist[:, 0] == dist[:, 1]
    assert tie.any() and np.all(idx[tie, 0] < idx[tie, 1])


def test_train_set_larger_than_one_index_window(vs, oracle):
    """Packed keys carry 20 index bits: more than 2^20 train rows forces several chunks with per-chunk offsets."""
    rng = np.random.default_rng(12)
    nt = (1 << 21) + 12345
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = t[rng.integers(0, nt, 96)].copy()
    q[:, 0] ^= 1  # distance 1 to its source row
    q[5] = t[nt - 1]            # exact hit in the very last row
    q[6] = t[(1 << 20)]         # exact hit at a chunk boundary
    t[(1 << 20) + 7] = t[3]     # duplicate across chunks: the lower index must come first
    q[7] = t[3]
    idx, dist = vs.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t, threads=0)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    assert idx[5, 0] == nt - 1 and idx[6, 0] == (1 << 20) and idx[7].tolist() == [3, (1 << 20) + 7]


def test_random_shapes_property(vs, oracle):
    rng = np.random.default_rng(99)
    for _ in range(25):
        nq = int(rng.integers(1, 3000))
        nt = int(rng.integers(2, 5000))
        bits = int(rng.integers(1, 9))  # low entropy -> many ties
        q = rng.integers(0, 1 << bits, (nq, 32)).astype(np.uint8)
        t = rng.integers(0, 1 << bits, (nt, 32)).astype(np.uint8)
        idx, dist = vs.hamming_knn2(q, t)
        oidx, odist = oracle.hamming_knn2(q, t, threads=0)
        assert np.array_equal(idx, oidx) and np.array_equal(dist, odist), (nq, nt, bits)
        ratio = float(rng.choice([0.5, 0.8, 0.9, 1.0]))
        a, b = vs.match_ratio(q, t, ratio), oracle.match_ratio(q, t, ratio)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_fused_fold_under_back_to_back_launches(vs, oracle):
    """The match is ONE launch: every (tile, chunk) workgroup publishes an 8-byte partial per query and the workgroup of the
    tile's last chunk folds them (self-validating words: launch epoch | second key | best key; round 3: a ticket).  Stale
    partial words from the previous launch would go unnoticed with repeated identical inputs, so different workloads alternate
    through the same scratch buffers, back to back without host synchronisation in between, and every result is checked -- in
    both train-row staging modes and over several launch geometries (1 chunk = no fold, few, many)."""
    import torch
    stream = torch.cuda.ExternalStream(vs.stream)
    sets = []
    for seed, (nq, nt) in enumerate([(3000, 5000), (3000, 5000), (2999, 4100)]):
        q, t = match_workload(nq, nt, n_dup=16, seed=70 + seed)
        sets.append((q, t) + oracle.hamming_knn2(q, t, threads=0))
    try:
        with torch.cuda.stream(stream):
            dev = [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t, _, _ in sets]
            outs = [(torch.empty((3000, 2), dtype=torch.int32, device="cuda"), torch.empty((3000, 2), dtype=torch.int32, device="cuda"))
                    for _ in range(24)]
            for tstage in (0, 1):
                vs.tune_match(tstage=tstage)
                for blocks in (0, 12, 300, 4000):
                    vs.tune_match(target_blocks=blocks)
                    for k, (oi, od) in enumerate(outs):        # 24 launches enqueued back to back
                        dq, dt = dev[k % 3]
                        vs.hamming_knn2_dev(dq.data_ptr(), dq.shape[0], dt.data_ptr(), dt.shape[0], oi.data_ptr(), od.data_ptr())
                    stream.synchronize()
                    for k, (oi, od) in enumerate(outs):
                        _, _, ridx, rdist = sets[k % 3]
                        n = ridx.shape[0]
                        assert np.array_equal(oi.cpu().numpy()[:n], ridx) and np.array_equal(od.cpu().numpy()[:n], rdist), (tstage, blocks, k)
    finally:
        vs.tune_match(target_blocks=0)
        vs.tune_match(tstage=1)


@pytest.mark.parametrize("tstage", [0, 1])
def test_both_train_staging_modes_bit_exact(vs, oracle, tstage):
    vs.tune_match(tstage=tstage)
    try:
        for nq, nt in [(1, 2), (257, 129), (1000, 3000), (5000, 63), (640, 65), (10000, 10000)]:
            q, t = match_workload(nq, nt, n_dup=min(16, nt // 4), seed=nq + 3 * nt)
            idx, dist = vs.hamming_knn2(q, t)
            oidx, odist = oracle.hamming_knn2(q, t, threads=0)
            assert np.array_equal(idx, oidx) and np.array_equal(dist, odist), (tstage, nq, nt)
    finally:
        vs.tune_match(tstage=1)


def test_partial_word_epochs_wrap_and_geometries_alternate(vs, oracle):
    """Round 4: the fold trusts a partial word iff it carries the launch's epoch (1 .. 63, cycling).  Every launch of one
    geometry rewrites every slot; a change of geometry (other tile / chunk counts on the same stream's scratch) clears the slots
    first.  200 launches back to back on one stream -- runs of one geometry long enough to wrap the epoch more than once,
    geometries alternating in between, a different workload almost every launch -- and every result checked."""
    import torch
    stream = torch.cuda.ExternalStream(vs.stream)
    shapes = [(1500, 2600), (1500, 2600), (700, 4100), (2600, 900), (1500, 2601)]
    sets = []
    for seed, (nq, nt) in enumerate(shapes):
        q, t = match_workload(nq, nt, n_dup=8, seed=170 + seed)
        sets.append((q, t) + oracle.hamming_knn2(q, t, threads=0))
    # 140 launches of geometry A / A' (same tile and chunk counts, different data), then switches every few launches
    order = [k % 2 for k in range(140)] + [2, 2, 0, 3, 3, 3, 1, 4, 4, 2] * 6
    with torch.cuda.stream(stream):
        dev = [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t, _, _ in sets]
        outs = [(torch.empty((2600, 2), dtype=torch.int32, device="cuda"), torch.empty((2600, 2), dtype=torch.int32, device="cuda"))
                for _ in order]
        for k, (oi, od) in zip(order, outs):
            dq, dt = dev[k]
            vs.hamming_knn2_dev(dq.data_ptr(), dq.shape[0], dt.data_ptr(), dt.shape[0], oi.data_ptr(), od.data_ptr())
        stream.synchronize()
    for n_launch, (k, (oi, od)) in enumerate(zip(order, outs)):
        _, _, ridx, rdist = sets[k]
        n = ridx.shape[0]
        assert np.array_equal(oi.cpu().numpy()[:n], ridx) and np.array_equal(od.cpu().numpy()[:n], rdist), (n_launch, k)
