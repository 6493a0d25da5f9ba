"""GPU parity: HIP gray / FAST-9 + NMS + cap / BRIEF-256 vs the CPU oracle, through the C ABI.  Bit-exact."""
import numpy as np
import pytest

from conftest import icl_frame
from visual_slam_amd.workloads import synthetic_frame

pytestmark = pytest.mark.gpu


def _eq_detect(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.shape == y.shape, (x.shape, y.shape)
        assert np.array_equal(x, y)


@pytest.mark.parametrize("shape", [(480, 640), (37, 53), (64, 64), (100, 301), (33, 1030)])
def test_gray(vs, oracle, shape):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, (*shape, 3), dtype=np.uint8)
    assert np.array_equal(vs.gray_mean3(img), oracle.gray_mean3(img))


@pytest.mark.parametrize("shape,seed", [((64, 64), 1), ((48, 100), 2), ((97, 61), 3), ((480, 640), 4), ((35, 257), 5),
                                        ((130, 1026), 6), ((7, 9), 7)])
def test_fast_detect_matches_oracle(vs, oracle, shape, seed):
    img = synthetic_frame(shape[1], shape[0], seed)[:, :, 0]
    for thr, border in ((20, 3), (35, 15), (10, 4), (1, 3), (254, 3)):
        for cap in (100000, 3000, 40, 7, 1, 0):
            _eq_detect(vs.fast9_detect(img, thr, border, cap), oracle.fast9_detect(img, thr, border, cap))


def test_fast_known_answers_on_gpu(vs):
    img = np.full((40, 64), 50, np.uint8)
    pos = [(10, 40 - 30), (20, 8), (20, 30), (30, 12), (40, 25)]
    for x, y in pos:
        img[y, x] = 150
    xy, sc = vs.fast9_detect(img, thr=20, border=3, max_kp=3)
    want = sorted(pos, key=lambda p: (p[1], p[0]))[:3]
    assert [tuple(p) for p in xy.astype(int)] == want and set(sc) == {99}
    img[:] = 100
    img[10, 10] = 121
    xy, sc = vs.fast9_detect(img, thr=20, border=3, max_kp=10)
    assert xy.tolist() == [[10.0, 10.0]] and sc.tolist() == [20]
    img[10, 10] = 120  # difference == thr is not a corner
    xy, _ = vs.fast9_detect(img, thr=20, border=3, max_kp=10)
    assert len(xy) == 0


def test_brief_matches_oracle(vs, oracle):
    img = synthetic_frame(200, 120, 5)[:, :, 0]
    rng = np.random.default_rng(4)
    n = 2500
    xy = np.stack([rng.uniform(-5, 205, n), rng.uniform(-5, 125, n)], 1).astype(np.float32)
    xy[:8] = [[15, 15], [184, 104], [14.4, 20], [14.5, 20], [15.5, 20], [184.5, 30], [185, 30], [40, 104.5]]
    d, keep = vs.brief256(img, xy)
    od, okeep = oracle.brief256(img, xy)
    assert np.array_equal(keep, okeep) and np.array_equal(d, od)
    d, keep = vs.brief256(img, np.zeros((0, 2), np.float32))
    assert d.shape == (0, 32)
    d, keep = vs.brief256(img, np.array([[0, 0], [1, 1]], np.float32))  # all dropped
    assert d.shape == (0, 32) and keep.shape == (0,)


@pytest.mark.parametrize("i", range(0, 20, 3))
def test_detect_describe_icl_frames(vs, oracle, i):
    bgr = icl_frame(i)
    _eq_detect(vs.detect_describe_bgr(bgr, 20, 3000), oracle.detect_describe_bgr(bgr, 20, 3000))


@pytest.mark.parametrize("cap", [3000, 500, 64, 1])
def test_detect_describe_synthetic_and_caps(vs, oracle, cap):
    bgr = synthetic_frame(640, 480, 2)
    a = vs.detect_describe_bgr(bgr, 20, cap)
    b = oracle.detect_describe_bgr(bgr, 20, cap)
    _eq_detect(a, b)
    assert len(a[0]) == min(cap, len(oracle.detect_describe_bgr(bgr, 20, 100000)[0]))
    # composition property: fused == gray -> detect(border 15) -> brief
    g = vs.gray_mean3(bgr)
    xy, sc = vs.fast9_detect(g, 20, 15, cap)
    d, keep = vs.brief256(g, xy)
    assert np.array_equal(xy, a[0]) and np.array_equal(sc, a[1]) and np.array_equal(d, a[2]) and len(keep) == len(xy)


def test_odd_sizes_and_strided_color(vs, oracle):
    rng = np.random.default_rng(11)
    for shape in ((61, 75), (33, 130), (90, 643)):
        g = synthetic_frame(shape[1], shape[0], shape[0])[:, :, 0]
        bgr = np.stack([g, np.roll(g, 1, 0), np.roll(g, 2, 1)], 2)
        bgr = np.ascontiguousarray(bgr)
        bgr[::7, ::5, 1] = rng.integers(0, 256, bgr[::7, ::5, 1].shape, dtype=np.uint8)
        _eq_detect(vs.detect_describe_bgr(bgr, 15, 3000), oracle.detect_describe_bgr(bgr, 15, 3000))


def test_flat_image_yields_nothing(vs):
    bgr = np.full((480, 640, 3), 77, np.uint8)
    xy, sc, d = vs.detect_describe_bgr(bgr, 20, 3000)
    assert xy.shape == (0, 2) and d.shape == (0, 32)


def test_deterministic(vs):
    bgr = icl_frame(5)
    a = vs.detect_describe_bgr(bgr, 20, 3000)
    b = vs.detect_describe_bgr(bgr, 20, 3000)
    _eq_detect(a, b)


def test_random_images_property(vs, oracle):
    """Random sizes, contrasts, thresholds, borders and caps: detection + description bit-exact."""
    rng = np.random.default_rng(123)
    for k in range(20):
        h, w = int(rng.integers(8, 200)), int(rng.integers(8, 700))
        base = synthetic_frame(w, h, int(rng.integers(0, 1000)))[:, :, 0].astype(np.int32)
        gain = float(rng.choice([0.2, 0.5, 1.0, 3.0]))
        g = np.clip((base - 128) * gain + 128 + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8)
        thr, border, cap = int(rng.integers(1, 80)), int(rng.integers(3, 20)), int(rng.choice([0, 1, 5, 50, 3000]))
        _eq_detect(vs.fast9_detect(g, thr, border, cap), oracle.fast9_detect(g, thr, border, cap))
        bgr = np.ascontiguousarray(np.stack([g, np.roll(g, 1, 1), g[::-1]], 2))
        _eq_detect(vs.detect_describe_bgr(bgr, thr, cap), oracle.detect_describe_bgr(bgr, thr, cap))


def test_saturated_and_extreme_images(vs, oracle):
    rng = np.random.default_rng(5)
    for img in (np.zeros((64, 96), np.uint8), np.full((64, 96), 255, np.uint8),
                (rng.integers(0, 2, (64, 96)) * 255).astype(np.uint8),          # salt and pepper: dense corners, scores 254
                np.tile(np.array([[0, 255], [255, 0]], np.uint8), (32, 48))):    # checkerboard
        for cap in (3000, 10):
            _eq_detect(vs.fast9_detect(img, 20, 3, cap), oracle.fast9_detect(img, 20, 3, cap))
        bgr = np.ascontiguousarray(np.repeat(img[:, :, None], 3, 2))
        _eq_detect(vs.detect_describe_bgr(bgr, 20, 3000), oracle.detect_describe_bgr(bgr, 20, 3000))


def test_frames_alternating_through_one_pinned_buffer(vs, oracle):
    """Different frames alternate through ONE pinned buffer (same addresses, new contents every call, DMA-ed from where they lie)
    for several hundred calls: no cache and no resident copy may ever serve rows of an earlier frame.  Plus assorted shapes, a
    pageable frame (staged) -- every result compared with the oracle's.  Since round 4 a pinned frame whose width is a multiple
    of 4 and whose bands all fit on the device at once is not copied at all: detect_band_kernel reads its own rows from the
    pinned memory, publishes them as gray rows behind a per-band flag (this frame's sequence number) and takes its halo from
    the neighbours' -- stale rows of the previous frame in the gray image, or a flag of an earlier frame, would show up here.
    Shapes that do not qualify (width not a multiple of 4, more bands than compute units) take the copy."""
    from visual_slam_amd.workloads import synthetic_frame
    frames = [icl_frame(i) for i in (0, 5, 10, 150)] + [synthetic_frame(640, 480, s) for s in (2, 7)]
    want = [oracle.detect_describe_bgr(f, 20, 3000) for f in frames]
    buf = vs.pin(np.zeros_like(frames[0]))
    for it in range(360):
        k = (it * 5 + it // 7) % len(frames)
        buf[...] = frames[k]
        xy, sc, desc = vs.detect_describe_bgr(buf, 20, 3000)
        oxy, osc, odesc = want[k]
        assert np.array_equal(xy, oxy) and np.array_equal(sc, osc) and np.array_equal(desc, odesc), (it, k)
    for w, h in ((640, 480), (321, 243), (100, 64), (1280, 720), (644, 480), (64, 16), (320, 600), (8, 8)):   # pinned, assorted shapes
        f = vs.pin(synthetic_frame(w, h, w + h))
        for _ in range(3):
            xy, sc, desc = vs.detect_describe_bgr(f, 20, 3000)
            oxy, osc, odesc = oracle.detect_describe_bgr(np.asarray(f), 20, 3000)
            assert np.array_equal(xy, oxy) and np.array_equal(sc, osc) and np.array_equal(desc, odesc), (w, h)
    f = synthetic_frame(640, 480, 11)                                 # pageable: staged, plain upload
    xy, sc, desc = vs.detect_describe_bgr(f, 20, 3000)
    oxy, osc, odesc = oracle.detect_describe_bgr(f, 20, 3000)
    assert np.array_equal(xy, oxy) and np.array_equal(desc, odesc)


@pytest.mark.gpu
@pytest.mark.parametrize("poison", [0x01, 0xFF, 0x7F])
def test_band_count_growing_inside_the_flag_buffer(oracle, poison):
    """Zero-copy detection keeps one flag word per band (the newest frame whose gray rows of the band are published).  The flag
    buffer is over-allocated: 240 bands (640 x 480) reserve room for 300.  A pinned frame with 250 bands (320 x 500) fits without
    a reallocation, so its last ten flags are whatever the ALLOCATION left there -- they must have been cleared then, or a word
    that happens to compare >= the frame's sequence number lets a neighbouring band read halo rows nobody has written.  Every
    device buffer of this context is poisoned on allocation (vs_debug_poison_alloc): 0x01010101 and 0x7F7F7F7F pass such a
    comparison, 0xFFFFFFFF never does (a band waiting on it runs out of its bounded wait).  The whole fused path runs under the
    same poison, so any other table assumed zero shows up here too."""
    from visual_slam_amd import Context
    from visual_slam_amd.workloads import synthetic_frame
    ctx = Context(0)
    ctx.debug_poison_alloc(poison)
    shapes = [(640, 480), (640, 480), (320, 500), (320, 500), (640, 512), (320, 500), (640, 480), (320, 598), (640, 480)]
    for k, (w, h) in enumerate(shapes):
        f = ctx.pin(synthetic_frame(w, h, 40 + k))
        for _ in range(2):
            xy, sc, desc = ctx.detect_describe_bgr(f, 20, 3000)
            oxy, osc, odesc = oracle.detect_describe_bgr(np.asarray(f), 20, 3000)
            assert np.array_equal(xy, oxy) and np.array_equal(sc, osc) and np.array_equal(desc, odesc), (poison, k, w, h)
    # the rest of the path on poisoned buffers: a match and a small BA
    from visual_slam_amd.workloads import ba_workload, match_workload
    q, t = match_workload(700, 900, n_dup=8, seed=5)
    idx, dist = ctx.hamming_knn2(q, t)
    oidx, odist = oracle.hamming_knn2(q, t)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)
    wk = ba_workload(n_cams=4, n_points=60, seed=2)
    args = (wk["poses"], wk["pose_fixed"], wk["points"], wk["point_fixed"], wk["obs_pose"], wk["obs_point"], wk["obs_uv"], wk["K"])
    g, o = ctx.ba_solve(*args, max_iterations=5), oracle.ba_solve(*args, max_iterations=5)
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(g["poses"], o["poses"])) < 1e-9
