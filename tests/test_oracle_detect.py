"""CPU: the C oracle's detector/descriptor against definition-level known answers and the NumPy twin."""
import os
import subprocess
import sys

import numpy as np
import pytest

import np_twin
from conftest import ROOT, icl_frame
from visual_slam_amd.workloads import synthetic_frame


def test_brief_pattern_header_is_reproducible(brief_pattern):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_brief_pattern.py")], capture_output=True,
                         text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "include", "vs_brief_pattern.h")).read()
    assert np.abs(brief_pattern).max() <= 13
    assert not np.any((brief_pattern[:, 0] == brief_pattern[:, 2]) & (brief_pattern[:, 1] == brief_pattern[:, 3]))


def test_gray_is_integer_mean(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    g = oracle.gray_mean3(img)
    assert np.array_equal(g, np_twin.gray_mean3(img))  # the reference's own expression (frame.py:11)
    assert np.array_equal(g, (img.astype(np.int32).sum(2) // 3).astype(np.uint8))


def _blank(v=100, n=21):
    return np.full((n, n), v, np.uint8)


def test_fast_known_answers(oracle):
    # (1) an isolated bright pixel: all 16 circle pixels are darker by 100 -> corner, score 99
    img = _blank(50)
    img[10, 10] = 150
    s = oracle.fast9_score_map(img, thr=20)
    assert s[10, 10] == 99 and np.count_nonzero(s) == 1
    # (2) exactly 9 contiguous brighter circle pixels -> corner; 8 -> not a corner
    for run, expect in ((9, True), (8, False)):
        img = _blank(100)
        for k in range(run):
            dx, dy = np_twin.CIRCLE[(5 + k) % 16]
            img[10 + dy, 10 + dx] = 160
        s = oracle.fast9_score_map(img, thr=20)
        assert (s[10, 10] > 0) == expect
        if expect:
            assert s[10, 10] == 59  # largest t with 160 > 100 + t
    # (3) threshold is strict: difference == thr is not a corner, thr + 1 is (score == thr)
    img = _blank(100)
    img[10, 10] = 120
    assert oracle.fast9_score_map(img, thr=20)[10, 10] == 0
    img[10, 10] = 121
    assert oracle.fast9_score_map(img, thr=20)[10, 10] == 20
    # (4) the arc may wrap around position 15 -> 0
    img = _blank(100)
    for k in range(9):
        dx, dy = np_twin.CIRCLE[(12 + k) % 16]
        img[10 + dy, 10 + dx] = 30
    assert oracle.fast9_score_map(img, thr=20)[10, 10] == 69
    # (5) border: nothing within `border` pixels of the edge
    img = _blank(50)
    img[3, 3] = 255
    img[2, 10] = 255
    s = oracle.fast9_score_map(img, thr=20, border=3)
    assert s[3, 3] > 0 and s[2, 10] == 0


def test_fast_nms_and_order(oracle):
    img = _blank(50, 40)
    img[10, 10] = 200
    img[10, 11] = 190  # neighbour with a lower score is suppressed
    img[20, 30] = 180
    img[20, 5] = 180
    xy, sc = oracle.fast9_detect(img, thr=20, border=3, max_kp=100)
    txy, tsc = np_twin.fast9_detect(img, thr=20, border=3, max_kp=100)
    assert np.array_equal(xy, txy) and np.array_equal(sc, tsc)
    assert [tuple(p) for p in xy.astype(int)] == sorted([tuple(p) for p in xy.astype(int)], key=lambda p: (p[1], p[0]))
    assert (10, 10) in [tuple(p) for p in xy.astype(int)] and (11, 10) not in [tuple(p) for p in xy.astype(int)]
    # equal scores side by side suppress each other (strictly-greater rule)
    img = _blank(50, 40)
    img[10, 10] = 200
    img[10, 12] = 200
    img[11, 11] = 200
    xy, _ = oracle.fast9_detect(img, thr=20, border=3, max_kp=100)
    txy, _ = np_twin.fast9_detect(img, thr=20, border=3, max_kp=100)
    assert np.array_equal(xy, txy)


@pytest.mark.parametrize("shape,seed", [((64, 64), 1), ((48, 100), 2), ((97, 61), 3)])
def test_fast_matches_twin_on_random_images(oracle, shape, seed):
    img = synthetic_frame(shape[1], shape[0], seed)[:, :, 0]
    for thr, border in ((20, 3), (35, 15), (10, 4)):
        assert np.array_equal(oracle.fast9_score_map(img, thr, border), np_twin.fast9_score_map(img, thr, border))
        for cap in (100000, 40, 7, 0):
            xy, sc = oracle.fast9_detect(img, thr, border, cap)
            txy, tsc = np_twin.fast9_detect(img, thr, border, cap)
            assert np.array_equal(xy, txy) and np.array_equal(sc, tsc), (thr, border, cap)


def test_fast_cap_keeps_strongest_ties_by_index(oracle):
    img = _blank(50, 64)
    # five identical corners (score 99), cap 3 -> the first three in row-major order
    pos = [(10, 40), (20, 8), (20, 30), (30, 12), (40, 50)]
    for x, y in pos:
        img[y, x] = 150
    xy, sc = oracle.fast9_detect(img, thr=20, border=3, max_kp=3)
    want = sorted(pos, key=lambda p: (p[1], p[0]))[:3]
    assert [tuple(p) for p in xy.astype(int)] == want and set(sc) == {99}
    # a stronger corner late in the image displaces a weak early one
    img[45, 45] = 255
    xy, sc = oracle.fast9_detect(img, thr=20, border=3, max_kp=3)
    assert (45, 45) in [tuple(p) for p in xy.astype(int)] and len(xy) == 3


def test_boxsum_and_brief_match_twin(oracle, brief_pattern):
    img = synthetic_frame(96, 80, 5)[:, :, 0]
    assert np.array_equal(oracle.boxsum5(img), np_twin.boxsum5(img))
    rng = np.random.default_rng(4)
    xy = np.stack([rng.uniform(0, 96, 200), rng.uniform(0, 80, 200)], 1).astype(np.float32)
    xy[:8] = [[15, 15], [80, 64], [14.4, 20], [14.5, 20], [15.5, 20], [80.5, 30], [81, 30], [40, 64.5]]  # edges + .5 ties
    d, keep = oracle.brief256(img, xy)
    td, tkeep = np_twin.brief256(img, xy, brief_pattern)
    assert np.array_equal(keep, tkeep) and np.array_equal(d, td)
    assert 0 in keep and 2 not in keep and 3 not in keep and 4 in keep  # 14.5 -> 14 (even) dropped, 15.5 -> 16 kept
    # known answer: on a horizontal ramp, bit k is 1 iff x1 < x2
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 3, (64, 1))
    d, _ = oracle.brief256(ramp, np.array([[32, 32]], np.float32))
    bits = np.unpackbits(d[0], bitorder="little")
    assert np.array_equal(bits, (brief_pattern[:, 0] < brief_pattern[:, 2]).astype(np.uint8))


def test_empty_inputs(oracle):
    img = _blank(100, 40)
    xy, sc = oracle.fast9_detect(img, 20, 3, 100)
    assert xy.shape == (0, 2)
    d, keep = oracle.brief256(img, np.zeros((0, 2), np.float32))
    assert d.shape == (0, 32) and keep.shape == (0,)


def test_detect_describe_on_icl_frame_matches_twin(oracle, brief_pattern):
    bgr = icl_frame(0)
    assert bgr.shape == (480, 640, 3)
    xy, sc, desc = oracle.detect_describe_bgr(bgr, thr=20, max_kp=3000)
    g = np_twin.gray_mean3(bgr)
    txy, tsc = np_twin.fast9_detect(g, 20, 15, 3000)
    td, tkeep = np_twin.brief256(g, txy, brief_pattern)
    assert len(xy) > 200
    assert np.array_equal(xy, txy) and np.array_equal(sc, tsc) and np.array_equal(desc, td) and len(tkeep) == len(txy)
