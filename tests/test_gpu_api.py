"""GPU: the src/v2-style Python API end to end on the HIP library, checked against the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, icl_frame
from visual_slam_amd.workloads import ICL_NUIM_K, ba_workload, match_workload

pytestmark = pytest.mark.gpu


def test_extractor_and_matcher_drop_in(vs, oracle):
    from visual_slam_amd.frame import FeatureExtractor, FeatureMatcher, Frame
    ex, ma = FeatureExtractor(context=vs), FeatureMatcher(context=vs)
    f0 = Frame(os.path.join(GOLDEN, "icl_nuim", "rgb", "0.png"), os.path.join(GOLDEN, "icl_nuim", "depth", "0.png"), 0)
    f1 = Frame(icl_frame(1), None, 1)
    assert f0.rgb.shape == (480, 640, 3) and np.array_equal(f0.rgb, icl_frame(0))
    kp0, ft0, rgb0 = f0.process_frame(ex)
    kp1, ft1, _ = f1.process_frame(ex)
    assert kp0.dtype == np.float32 and kp0.shape[1] == 2 and ft0.shape == (len(kp0), 32) and rgb0 is f0.rgb
    oxy, _, odesc = oracle.detect_describe_bgr(icl_frame(0), 20, 3000)
    assert np.array_equal(kp0, oxy) and np.array_equal(ft0, odesc)
    matches, p1, d1, p2, d2 = ma.match_features(kp0, ft0, kp1, ft1)
    oq, ot, od = oracle.match_ratio(ft0, ft1, 0.8)
    assert [m[0].queryIdx for m in matches] == oq.tolist() and [m[0].trainIdx for m in matches] == ot.tolist()
    assert [m[0].distance for m in matches] == od.astype(float).tolist()
    assert np.array_equal(p1, kp0[oq]) and np.array_equal(p2, kp1[ot]) and np.array_equal(d1, ft0[oq]) and np.array_equal(d2, ft1[ot])
    assert len(matches) > 100
    assert f0.depth.shape == (480, 640, 3)
    with pytest.raises(ValueError):  # the reference cannot unpack (m, n) with a single train descriptor
        ma.match_features(kp0, ft0, kp1[:1], ft1[:1])
    m0 = ma.match_features(kp0[:0], ft0[:0], kp1, ft1)
    assert len(m0[0]) == 0 and m0[1].shape == (0, 2)


def test_bundle_adjustment_class_on_gpu(vs, oracle):
    from test_host_api import _scene_map
    from visual_slam_amd.LocalBA import BundleAdjustment, Camera
    w = ba_workload(n_cams=5, n_points=120, seed=33)
    m_gpu, m_cpu = _scene_map(w), _scene_map(w)
    BundleAdjustment(Camera(*ICL_NUIM_K), context=vs).localBundleAdjustement(m_gpu)
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).localBundleAdjustement(m_cpu)
    for i in range(5):
        a, b = m_gpu.GetFrame(i).GetPose(), m_cpu.GetFrame(i).GetPose()
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-8
    assert np.allclose(m_gpu.GetAll3DPoints(), m_cpu.GetAll3DPoints(), atol=1e-8)
    m_gpu, m_cpu = _scene_map(w), _scene_map(w)
    BundleAdjustment(Camera(*ICL_NUIM_K), context=vs).motionOnlyBundleAdjustement(m_gpu)
    BundleAdjustment(Camera(*ICL_NUIM_K), solver=oracle.ba_solve).motionOnlyBundleAdjustement(m_cpu)
    for i in range(5):
        a, b = m_gpu.GetFrame(i).GetPose(), m_cpu.GetFrame(i).GetPose()
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-8


def test_tracking_harness_gpu_equals_oracle(vs, oracle):
    """BASELINE.json configs[0]: the first 20 frames end to end (here: traj3), GPU vs CPU path."""
    from visual_slam_amd import harness

    def odetect(bgr):
        xy, _, desc = oracle.detect_describe_bgr(bgr, 20, 3000)
        return xy, desc

    def omatch(q, t):
        mq, mt, _ = oracle.match_ratio(q, t, 0.8)
        return mq, mt

    def oba(*p):
        return oracle.ba_solve(*p, huber_delta=harness.HUBER, max_iterations=10)

    frames, depth0 = harness.load_sequence(20)
    def opnp(obj, img, K4, pose0, seed=0):
        return oracle.pnp_ransac(obj, img, K4, pose0, seed=seed)

    for gp, cp in ((None, None), (harness.gpu_pnp(vs), opnp)):   # without and with the PnP-RANSAC stage (main.py:196)
        gposes, _, gn = harness.track_sequence(*harness.gpu_callables(vs), frames, depth0, pnp=gp)
        cposes, _, cn = harness.track_sequence(odetect, omatch, oba, frames, depth0, pnp=cp)
        assert gn == cn and min(gn) > 100
        rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(gposes, cposes))
        assert rel < 1e-4, rel
        step = np.linalg.norm(np.diff(gposes[:, :3, 3], axis=0), axis=1)
        assert step.max() < 0.05  # consecutive ICL-NUIM frames are millimetres apart: the tracker must not jump
    # the device-resident session (map uploaded once, one image upload per frame) runs the same kernels
    rposes, _, rn = harness.track_sequence_resident(vs, frames, depth0)
    assert rn == gn
    # (the session keeps camera records, the array path converts 4x4 poses to quaternions every call: rounding-level
    # differences that an accept/reject decision of the LM at convergence can amplify to ~1e-9)
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(rposes, gposes)) < 1e-7
    # ... and pipelining frame k+1's upload / detection / match with frame k's PnP + BA changes nothing
    pposes, _, pn = harness.track_sequence_resident(vs, frames, depth0, pipelined=True)
    assert pn == rn and np.array_equal(pposes, rposes)
    # the class-API period (Frame / Map / FeatureMatcher / solvePnPRansac / BundleAdjustment) tracks the same poses
    aposes, _ = harness.track_sequence_api(frames, depth0, context=vs)
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(aposes, gposes)) < 1e-6


def test_sharded_matcher_world1_hip_path(vs, oracle):
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    q, t = match_workload(3000, 2000, seed=8)
    m = ShardedMatcher()
    with torch.cuda.stream(m.torch_stream()):
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        idx, dist = m.knn2(dq, dt)
        idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    oidx, odist = oracle.hamming_knn2(q, t)
    assert np.array_equal(idx, oidx) and np.array_equal(dist, odist)


def test_sharded_matcher_on_the_callers_stream(vs, oracle):
    """Round-1 advisor finding: inputs produced on, and results consumed from, torch's CURRENT stream (not the library's)
    must be ordered by the matcher itself.  The producer here is a long chain of device ops on the default stream that
    finishes writing q just before the match is submitted; a missing wait would read the buffer half-written."""
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    m = ShardedMatcher()
    for rep in range(5):
        q, t = match_workload(6000, 3000, seed=40 + rep)
        junk = torch.zeros((6000, 32), dtype=torch.uint8, device="cuda")
        dq = torch.empty((6000, 32), dtype=torch.uint8, device="cuda")
        dt = torch.from_numpy(t).cuda()
        src = torch.from_numpy(q).cuda()
        torch.cuda.synchronize()
        big = torch.ones((4096, 4096), device="cuda")
        for _ in range(6):                       # keeps the default stream busy for a while ...
            big = big @ big * 1e-4
        dq.copy_(junk)
        dq.copy_(src)                            # ... and only then produces the real queries
        idx, dist = m.knn2(dq, dt)               # default stream is current: the matcher must wait for the copy
        got = (idx.clone(), dist.clone())        # consumed on the default stream: must wait for the kernel
        torch.cuda.synchronize()
        oidx, odist = oracle.hamming_knn2(q, t, threads=0)
        assert np.array_equal(got[0].cpu().numpy(), oidx) and np.array_equal(got[1].cpu().numpy(), odist), rep
        assert float(big.sum()) == float(big.sum())


def test_sharded_step_through_the_c_abi_with_a_collective_at_world_1(vs, oracle):
    """vs_hamming_knn2_sharded_dev with a real RCCL communicator of one rank: kernel into the gather slot + in-place
    ncclAllGather on RCCL's own stream + done event; rotating buffers through submit / collect as bench.py drives them."""
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    dist = _one_rank_group(vs)
    try:
        m = ShardedMatcher(force_collective=True)
        work = [match_workload(2000 + 37 * k, 1500, seed=60 + k) for k in range(6)]
        dev = [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t in work]
        torch.cuda.synchronize()
        results, pending = [], []
        for k, (dq, dt) in enumerate(dev):
            ticket = m.submit(dq, dt, dq.shape[0])
            if pending:
                i, d = m.collect(pending.pop())
                results.append((i.clone(), d.clone()))
            pending.append(ticket)
        i, d = m.collect(pending.pop())
        results.append((i.clone(), d.clone()))
        torch.cuda.synchronize()
        assert m._rccl is not None, "the direct RCCL path did not initialise"
        for (q, t), (i, d) in zip(work, results):
            oi, od = oracle.hamming_knn2(q, t, threads=0)
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od)
        m.close()
    finally:
        dist.destroy_process_group()

def _one_rank_group(vs):
    import os
    import socket
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", vs.device))
    return dist


@pytest.mark.parametrize("in_flight,collective", [(2, True), (3, True), (1, True), (2, False)])
def test_plan_keeps_steps_in_flight_without_racing_its_own_buffers(vs, oracle, in_flight, collective):
    """Round-2 verdict / advisor: ShardedMatcher.plan(in_flight >= 2) runs every slot on a stream of its own.  The step
    bench.py --gpus N drives must then order, per slot, the kernel behind (a) the slot's previous in-place all-gather,
    (b) the consumer of the slot's previous results and (c) the producer of the new inputs -- all three on other streams.
    Here: a real one-rank RCCL communicator (the direct path), 14 steps, a different workload every step written into
    rotating input buffers on the library's stream right after a long-running producer chain, and a slow consumer (a
    matmul chain, then the copy of the results) after every collect.  Every collected result is checked."""
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    dist = _one_rank_group(vs) if collective else None
    try:
        m = ShardedMatcher(force_collective=collective)
        stream = m.torch_stream()
        nq, nt, steps = 6000, 3000, 14
        work = [match_workload(nq, nt, n_dup=8, seed=300 + k) for k in range(steps)]
        with torch.cuda.stream(stream):
            src = [(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()) for q, t in work]
            nbuf = max(2, in_flight) + 1   # inputs stay untouched until their step is collected
            qbuf = [torch.zeros((nq, 32), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
            tbuf = [torch.zeros((nt, 32), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
            big = torch.ones((2048, 2048), device="cuda")
            plan = m.plan(qbuf[0], tbuf[0], nq, in_flight=in_flight)
            if collective:
                assert plan.direct and m._rccl is not None, "the direct RCCL path did not initialise"
            assert len({s.cuda_stream for s in plan.streams}) == (1 if in_flight == 1 else max(2, in_flight))
            pending, results = [], []

            def consume(slot):
                nonlocal big
                i, d = plan.collect(slot)
                for _ in range(3):                      # slow consumer on the library's stream ...
                    big = big @ big * 1e-4
                results.append((i.clone(), d.clone()))  # ... that reads the slot's buffer late

            for k in range(steps):
                for _ in range(2):                      # slow producer: the inputs are written late, too
                    big = big @ big * 1e-4
                qbuf[k % nbuf].copy_(src[k][0])
                tbuf[k % nbuf].copy_(src[k][1])
                pending.append(plan.submit(qbuf[k % nbuf], tbuf[k % nbuf]))
                while len(pending) > max(1, in_flight - 1):
                    consume(pending.pop(0))
            while pending:
                consume(pending.pop(0))
            stream.synchronize()
        torch.cuda.synchronize()
        assert len(results) == steps
        for k, ((q, t), (i, d)) in enumerate(zip(work, results)):
            oi, od = oracle.hamming_knn2(q, t, threads=0)
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od), (in_flight, k)
        m.close()
    finally:
        if dist is not None:
            dist.destroy_process_group()


@pytest.mark.parametrize("collective,in_flight", [(True, 2), (False, 2), (True, 1)])
def test_static_input_plan_runs_the_exchange_on_the_librarys_stream(vs, oracle, collective, in_flight):
    """Round 4: plan(static_inputs=True, buffers=4) is what bench.py drives with a collective -- the step's kernel is not
    ordered behind the library's stream, more buffer sets rotate than streams run, and the in-place all-gather runs on the
    library's own stream (a stream made later may share a hardware queue with a compute stream).  in_flight = 1 with a collective
    (round 5: the mode of bench.py's `value` at N > 1): the kernels one at a time on the context's first auxiliary stream, the
    all-gathers on its second, the library's stream carrying only the consumers' waits.  The inputs never change,
    so every slot's buffer is poisoned after its results were read: a step that returned before its kernel and exchange had
    rewritten the slot would show the poison."""
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    dist = _one_rank_group(vs) if collective else None
    try:
        m = ShardedMatcher(force_collective=collective)
        stream = m.torch_stream()
        nq, nt, steps = 5000, 4000, 13
        q, t = match_workload(nq, nt, n_dup=8, seed=411)
        oi, od = oracle.hamming_knn2(q, t, threads=0)
        with torch.cuda.stream(stream):
            dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
            stream.synchronize()
            plan = m.plan(dq, dt, nq, in_flight=in_flight, buffers=4, static_inputs=True)
            assert plan.nslots == 4 and len({s.cuda_stream for s in plan.streams}) == in_flight
            if collective and in_flight == 2:
                assert plan.direct and plan.args[0][11] == stream.cuda_stream
            elif collective:
                comp, comm = plan.args[0][10], plan.args[0][11]
                assert plan.direct and comp == vs.aux_stream(0) and comm == vs.aux_stream(1) and stream.cuda_stream not in (comp, comm)
            with pytest.raises(ValueError):
                plan.submit(q=dq)
            pending, results = [], []

            def consume(slot):
                i, d = plan.collect(slot)
                results.append((i.clone(), d.clone()))
                for buf in plan.bufs[slot]:
                    if buf is not None:
                        buf.fill_(-1)
                stream.synchronize()   # the contract of static_inputs: done with a slot's results before its next submit

            for k in range(steps):
                pending.append(plan.submit())
                while len(pending) > 3:
                    consume(pending.pop(0))
            while pending:
                consume(pending.pop(0))
        torch.cuda.synchronize()
        assert len(results) == steps
        for k, (i, d) in enumerate(results):
            assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od), k
        m.close()
    finally:
        if dist is not None:
            dist.destroy_process_group()


def test_descriptor_cache_never_serves_stale_data(vs, oracle):
    """The host matcher keeps device copies keyed by (address, n) and verified byte-for-byte against a host shadow of the
    uploaded bytes: in-place edits and buffer re-use must still give the oracle's answer (knnMatch never caches,
    reference src/v2/frame.py:23)."""
    q, t = match_workload(800, 700, seed=21)
    a = vs.hamming_knn2(q, t)
    assert np.array_equal(a[0], oracle.hamming_knn2(q, t)[0])
    t[5] ^= 0xFF                      # same address, different content
    q[::7] = np.roll(q[::7], 3, axis=1)
    b = vs.hamming_knn2(q, t)
    o = oracle.hamming_knn2(q, t)
    assert np.array_equal(b[0], o[0]) and np.array_equal(b[1], o[1])
    for seed in range(10):            # more sets than cache slots, all through the same two host buffers
        q2, t2 = match_workload(800, 700, seed=100 + seed)
        q[...] = q2
        t[...] = t2
        mq, mt, md = vs.match_ratio(q, t, 0.8)
        oq, ot, od = oracle.match_ratio(q2, t2, 0.8)
        assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)


def test_descriptor_cache_two_byte_edits(vs, oracle):
    """Round-1 regression: the cache key used to be a multiply-xor fingerprint in which the top bit of two words of one
    lane cancelled, so `t[k,7] ^= 0x80; t[k+1,7] ^= 0x80` (same buffer, edited in place) was served from the stale device
    copy.  The cache now compares bytes; every in-place edit must be seen."""
    q, t = match_workload(700, 700, seed=33)
    # queries sit next to their train partner, so flipping a train bit changes some distance that is reported
    for k in (5, 6, 100, 698):
        q[k] = t[k]
    assert np.array_equal(vs.hamming_knn2(q, t)[1], oracle.hamming_knn2(q, t)[1])
    for col in (7, 15, 23, 31):       # the four byte positions whose MSB was linear in the old fingerprint
        t[5, col] ^= 0x80
        t[6, col] ^= 0x80
        g, o = vs.hamming_knn2(q, t), oracle.hamming_knn2(q, t)
        assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]), col
        assert g[1][5, 0] > 0 and g[1][6, 0] > 0      # the edit is visible in the result (it would be 0 from a stale copy)
        t[5, col] ^= 0x80
        t[6, col] ^= 0x80
    rng = np.random.default_rng(7)
    for trial in range(300):          # random two-byte edits of either operand, alternating entry points
        arr = t if trial & 1 else q
        for _ in range(2):
            arr[rng.integers(arr.shape[0]), rng.integers(32)] ^= np.uint8(1 << rng.integers(8))
        if trial % 3 == 0:
            mq, mt, md = vs.match_ratio(q, t, 0.8)
            oq, ot, od = oracle.match_ratio(q, t, 0.8)
            assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od), trial
        else:
            g, o = vs.hamming_knn2(q, t), oracle.hamming_knn2(q, t)
            assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]), trial
    # the detector's own output array is adopted without an upload; editing it in place afterwards must be seen too
    from visual_slam_amd.workloads import synthetic_frame
    xy, sc, desc = vs.detect_describe_bgr(synthetic_frame(320, 240, 5), 20, 1000)
    d2 = desc.copy()
    assert np.array_equal(vs.hamming_knn2(desc, t)[1], oracle.hamming_knn2(d2, t)[1])
    desc[3, 7] ^= 0x80
    desc[4, 7] ^= 0x80
    d2[3, 7] ^= 0x80
    d2[4, 7] ^= 0x80
    g, o = vs.hamming_knn2(desc, t), oracle.hamming_knn2(d2, t)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])


def test_pinned_frames_and_resident_descriptors(vs, oracle):
    bgr = icl_frame(2)
    pinned = vs.pin(bgr)
    a = vs.detect_describe_bgr(pinned, 20, 3000)
    b = oracle.detect_describe_bgr(bgr, 20, 3000)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    a2 = vs.detect_describe_bgr(icl_frame(3), 20, 3000)
    mq, mt, md = vs.match_ratio(a[2], a2[2], 0.8)   # both descriptor sets are already resident on the device
    oq, ot, od = oracle.match_ratio(b[2], oracle.detect_describe_bgr(icl_frame(3), 20, 3000)[2], 0.8)
    assert np.array_equal(mq, oq) and np.array_equal(mt, ot) and np.array_equal(md, od)


def test_tracking_session_details(vs, oracle):
    """vs_track_*: optional outputs equal the stand-alone entry points, no-PnP mode, capacity and misuse errors."""
    from visual_slam_amd import harness
    from visual_slam_amd.context import VsError
    frames, depth0 = harness.load_sequence(4)
    xy0, _, d0 = vs.detect_describe_bgr(frames[0], 20, 3000)
    X = harness.backproject(xy0, depth0)
    with pytest.raises(VsError):
        vs.track_end() or vs.track_frame(frames[1])           # no period open
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=2, pnp_iterations=0)
    r = vs.track_frame(frames[1], want_keypoints=True)
    xy1, _, d1 = vs.detect_describe_bgr(frames[1], 20, 3000)
    mq, mt, _ = vs.match_ratio(d0, d1, 0.8)
    assert np.array_equal(r["xy"], xy1) and np.array_equal(r["desc"], d1) and r["n_keypoints"] == len(xy1)
    assert np.array_equal(r["match_q"], mq) and np.array_equal(r["match_t"], mt) and not r["pnp_found"]
    # without PnP the session is motion-only BA from the previous pose: identical to the stand-alone solve
    lm = harness.LocalMapArrays(X)
    lm.add_frame(np.eye(4), mq, xy1[mt])
    ref = vs.ba_solve(*lm.problem(), huber_delta=harness.HUBER, max_iterations=10)
    assert r["poses"].shape == (2, 4, 4) and np.abs(r["poses"][1] - ref["poses"][1]).max() < 1e-12
    vs.track_frame(frames[2])
    with pytest.raises(VsError):
        vs.track_frame(frames[3])                               # max_frames = 2
    vs.track_end()
    # pipelined entry point: result of frame k arrives with the submission of frame k+1; flush with None; optional outputs
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=3, pnp_iterations=0)
    assert vs.track_frame_pipelined(frames[1], want_keypoints=True) is None
    with pytest.raises(VsError):
        vs.track_frame(frames[2])                               # a pipelined frame is pending
    p1 = vs.track_frame_pipelined(frames[2], want_keypoints=True)
    assert np.array_equal(p1["xy"], xy1) and np.array_equal(p1["match_q"], mq) and np.array_equal(p1["match_t"], mt)
    assert np.abs(p1["poses"][1] - ref["poses"][1]).max() < 1e-12
    p2 = vs.track_frame_pipelined(None)
    assert p2["poses"].shape == (3, 4, 4) and vs.track_frame_pipelined(None) is None
    vs.track_end()


def test_tracking_session_survives_frames_without_features(vs):
    """A black frame has no keypoints: no matches, PnP finds nothing, the pose stays at the previous one and the period
    continues with the next real frame (the reference would raise in knnMatch; the session reports 0 matches)."""
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(3)
    xy0, _, d0 = vs.detect_describe_bgr(frames[0], 20, 3000)
    vs.track_begin(harness.backproject(xy0, depth0), d0, np.eye(4), ICL_NUIM_K, max_frames=4)
    black = np.zeros_like(frames[0])
    r = vs.track_frame(black, want_keypoints=True)
    assert r["n_keypoints"] == 0 and r["n_matches"] == 0 and not r["pnp_found"]
    assert np.allclose(r["poses"][1], np.eye(4), atol=1e-12)
    r1 = vs.track_frame(frames[1])
    assert r1["n_matches"] > 100 and r1["pnp_found"] and r1["poses"].shape == (3, 4, 4)
    assert np.allclose(r1["poses"][1], np.eye(4), atol=1e-12)            # the empty camera is untouched by the BA
    assert np.linalg.norm(r1["poses"][2][:3, 3]) < 0.05 and np.isfinite(r1["poses"]).all()
    r2 = vs.track_frame(black)
    assert r2["n_matches"] == 0 and np.abs(r2["poses"][3] - r2["poses"][2]).max() < 1e-9
    vs.track_end()


def test_argument_errors_of_the_widened_entry_points(vs):
    """Status codes instead of crashes: mismatched lengths, negative counts, capacities (C ABI returns VS_EINVAL / VS_ENOMEM,
    the Python layer raises VsError / ValueError)."""
    from visual_slam_amd.context import VsError
    x = np.zeros((10, 2))
    with pytest.raises(ValueError):
        vs.essential_ransac(x, x[:5], 1e-3)
    with pytest.raises(ValueError):
        vs.recover_pose(np.eye(3), x, x[:5])
    with pytest.raises(VsError):
        vs.essential_ransac(x, x, 1e-3, max_iters=-1)
    with pytest.raises(VsError):
        vs.pnp_ransac(np.zeros((10, 3)), x, ICL_NUIM_K, np.eye(4), refine_iters=-1)
    with pytest.raises(ValueError):
        vs.track_begin(np.zeros((10, 3)), np.zeros((9, 32), np.uint8), np.eye(4), ICL_NUIM_K)
    with pytest.raises(VsError):
        vs.track_begin(np.zeros((10, 3)), np.zeros((10, 32), np.uint8), np.eye(4), ICL_NUIM_K, max_frames=0)
    with pytest.raises(VsError):
        vs.track_begin(np.zeros((10, 3)), np.zeros((10, 32), np.uint8), np.eye(4), ICL_NUIM_K, pnp_iterations=100000)
    # degenerate but legal inputs give "not found", not an error
    assert not vs.essential_ransac(np.zeros((20, 2)), np.zeros((20, 2)), 1e-3)["found"]          # all points coincide
    r = vs.pnp_ransac(np.zeros((20, 3)), np.zeros((20, 2)), ICL_NUIM_K, np.eye(4))              # points at the centre
    assert isinstance(r["found"], bool) and np.isfinite(r["pose"]).all() or not r["found"]


def test_tracking_session_on_an_odd_sized_image(vs):
    """Width 637: the rows are not a multiple of four bytes, so the session's upload goes through the pitched copy."""
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(3)
    crop = [np.ascontiguousarray(f[3:470, 2:639]) for f in frames]          # 467 x 637
    xy0, _, d0 = vs.detect_describe_bgr(crop[0], 20, 3000)
    X = harness.backproject(xy0 + np.array([2, 3], np.float32), depth0)
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=2, pnp_iterations=0)
    r = vs.track_frame(crop[1], want_keypoints=True)
    xy1, _, d1 = vs.detect_describe_bgr(crop[1], 20, 3000)
    mq, mt, _ = vs.match_ratio(d0, d1, 0.8)
    assert np.array_equal(r["xy"], xy1) and np.array_equal(r["desc"], d1)
    assert np.array_equal(r["match_q"], mq) and np.array_equal(r["match_t"], mt) and len(mq) > 100
    vs.track_end()


def test_class_api_keeps_the_tracking_period_resident(vs, oracle):
    """SURVEY 8f rank 1 behind the class API: the unmodified call sequence of the reference's tracking loop
    (main.py:181-214 -- Frame.process_frame, Map.GetImagePointsWithFrameID, FeatureMatcher.match_features,
    solvePnPRansac, Map.AddParentAndPose / AddPointToFrameCorrespondences, BundleAdjustment.motionOnlyBundleAdjustement)
    must route the per-frame BA to the device-resident period (only the new frame travels) and give the poses of the
    path that rebuilds and uploads the whole problem every frame, to 1e-9."""
    from visual_slam_amd import harness
    from visual_slam_amd.map import Map
    frames, depth0 = harness.load_sequence(12)
    Map.use_device_mirror = False
    try:
        ref_poses, _ = harness.track_sequence_api(frames, depth0, context=vs)
    finally:
        Map.use_device_mirror = True
    # how the frames reach the resident period: the first one host-fed (vs_track_push_frame: the period does not exist before
    # the first motionOnlyBundleAdjustement), every later one detected, matched and PnP-ed inside the period (vs_track_front /
    # vs_track_back_begin) and only collected by the BA call (vs_track_back_end); the period is begun once, never rebuilt
    calls = {"push": [], "front": 0, "back_begin": 0, "back_end": 0, "begin": 0}
    orig = {n: getattr(vs, n) for n in ("track_push_frame", "track_front", "track_back_begin", "track_back_end", "track_begin")}

    def spy(name, key, record=None):
        def f(*a, **k):
            if record is not None:
                calls[key].append(record(a))
            else:
                calls[key] += 1
            return orig[name](*a, **k)
        return f
    vs.track_push_frame = spy("track_push_frame", "push", lambda a: len(a[0]))
    vs.track_front = spy("track_front", "front")
    vs.track_back_begin = spy("track_back_begin", "back_begin")
    vs.track_back_end = spy("track_back_end", "back_end")
    vs.track_begin = spy("track_begin", "begin")
    try:
        poses, _ = harness.track_sequence_api(frames, depth0, context=vs)
    finally:
        for n in orig:
            delattr(vs, n)
    assert calls["begin"] == 1 and len(calls["push"]) == 1 and min(calls["push"]) > 50, calls
    assert calls["back_begin"] == calls["back_end"] == len(frames) - 2, calls
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(poses, ref_poses))
    assert rel <= 1e-9, rel
    # edits behind the mirror's back restart the period instead of using stale device state
    from visual_slam_amd.LocalBA import BundleAdjustment, Camera
    from visual_slam_amd.frame import FeatureExtractor, FeatureMatcher, Frame
    from visual_slam_amd.point import Point
    key = Frame(frames[0], None, 0)
    key.AddPose(np.eye(4))
    key.SetAsKeyFrame()
    kp0, ft0, _ = key.process_frame(FeatureExtractor(context=vs))
    maps = []
    for use in (True, False):
        Map.use_device_mirror = use
        m = Map()
        m.AddFrame(0, key)
        for i, (X, uv, d) in enumerate(zip(harness.backproject(kp0, depth0), kp0, ft0)):
            pt = Point(location=X, id=i + 1)
            pt.AddFrame(frame=key, uv=uv, descriptor=d)
            m.AddPoint3D(point_id=i + 1, point_3d=pt)
        for k in (1, 2, 3):
            cur = Frame(frames[k], None, k)
            kp, ft, _ = cur.process_frame(FeatureExtractor(context=vs))
            kpp, ftp, xyz, ids = m.GetImagePointsWithFrameID(0)
            matches, _, _, cp, cf = FeatureMatcher(context=vs).match_features(kpp, ftp, kp, ft)
            m.AddParentAndPose(parent_id=k - 1, frame_id=k, frame_obj=cur, rel_pose_trans=np.eye(4), pose=np.eye(4))
            m.AddPointToFrameCorrespondences(ids[matches.query_idx], cp, cf, cur)
            if k == 2:
                m.UpdatePoint3D(np.asarray(m.GetPoint(7).Get3dPoint()) + 0.01, 7)   # geometry edit between two solves
                m.UpdatePose(np.asarray(m.GetFrame(1).GetPose()) @ np.eye(4), 1)    # pose object replaced (same value)
            BundleAdjustment(Camera(*ICL_NUIM_K), context=vs).motionOnlyBundleAdjustement(m)
        maps.append(np.stack([m.GetFrame(k).GetPose() for k in range(4)]))
    Map.use_device_mirror = True
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(*maps)) <= 1e-9


def test_resident_tracking_one_launch_solve_equals_launch_per_step(vs):
    """The resident tracking period with the motion-only BA as one launch per frame against one launch per LM step:
    identical poses, frame by frame and pipelined."""
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(12)
    try:
        vs.tune_ba(motion_variant=1)
        ref, _, _ = harness.track_sequence_resident(vs, frames, depth0)
        vs.tune_ba(motion_variant=0)
        got, _, _ = harness.track_sequence_resident(vs, frames, depth0)
        piped, _, _ = harness.track_sequence_resident(vs, frames, depth0, pipelined=True)
    finally:
        vs.tune_ba(motion_variant=0)
    assert np.array_equal(ref, got) and np.array_equal(ref, piped)


def test_resident_tracking_survives_frames_without_key_points(vs, oracle):
    """The resident front half never brings the key-point count to the host: the matcher is launched for max_kp train rows
    and reads the count on the device.  A blank frame (no corner anywhere: zero key points) and a frame with a single
    corner (one key point: no second neighbour) must give zero matches, leave the pose at the previous one, and the period
    must go on with the next real frame exactly as a period that never saw them would from the same state."""
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(4)
    xy, _, desc = oracle.detect_describe_bgr(frames[0], 20, 3000)
    X = harness.backproject(xy, depth0)
    blank = np.full_like(frames[0], 90)
    one = blank.copy()
    one[200:216, 300:316] = 255  # a bright square: at most a handful of corners, all in one place
    vs.track_begin(X, desc, np.eye(4), ICL_NUIM_K, max_frames=8)
    try:
        r1 = vs.track_frame(frames[1], seed=1)
        rb = vs.track_frame(blank, seed=2, want_keypoints=True)
        assert rb["n_matches"] == 0 and not rb["pnp_found"] and rb["xy"].shape[0] == 0
        assert np.allclose(rb["poses"][2], r1["poses"][1], rtol=0, atol=1e-6)  # nothing to go on: the start pose (frame 1's) stays
        ro = vs.track_frame(one, seed=3, want_keypoints=True)
        assert ro["n_matches"] <= ro["xy"].shape[0] and not ro["pnp_found"]
        r2 = vs.track_frame(frames[2], seed=4)
        assert r2["n_matches"] > 100 and r2["pnp_found"]
        assert np.all(np.isfinite(r2["poses"]))
        assert np.allclose(r2["poses"][1], r1["poses"][1], atol=5e-3)  # the first frame's pose only moves within the BA's reach
    finally:
        vs.track_end()


def test_class_api_speculation_falls_back_when_the_caller_deviates(vs, oracle):
    """The class API runs a frame's front half and PnP inside the resident period AHEAD of the calls that ask for them
    (vs_track_front in process_frame, vs_track_back_begin in solvePnPRansac).  That is only valid while the caller passes
    on exactly what the device used.  Here the caller deviates -- drops matches before PnP, ignores the PnP pose, edits the
    match list between PnP and the BA, skips PnP altogether -- and every time the result must be that of the path that
    rebuilds and uploads the whole problem (Map.use_device_mirror = False) driven the same way, to 1e-9."""
    from visual_slam_amd import harness
    from visual_slam_amd import helper_functions as hf
    from visual_slam_amd.LocalBA import BundleAdjustment, Camera
    from visual_slam_amd.frame import FeatureExtractor, FeatureMatcher, Frame
    from visual_slam_amd.map import Map
    from visual_slam_amd.point import Point
    frames, depth0 = harness.load_sequence(9)
    fx, fy, cx, cy = ICL_NUIM_K
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])

    def drive():
        extractor, matcher, camera = FeatureExtractor(context=vs), FeatureMatcher(context=vs), Camera(*ICL_NUIM_K)
        key = Frame(frames[0], None, 0)
        key.AddPose(np.eye(4))
        key.SetAsKeyFrame()
        kp0, ft0, _ = key.process_frame(extractor)
        m = Map()
        m.AddFrame(0, key)
        for i, (X, uv, d) in enumerate(zip(harness.backproject(kp0, depth0), kp0, ft0)):
            pt = Point(location=X, id=i + 1)
            pt.AddFrame(frame=key, uv=uv, descriptor=d)
            m.AddPoint3D(point_id=i + 1, point_3d=pt)
        for k in range(1, len(frames)):
            cur = Frame(frames[k], None, k)
            kp_cur, ft_cur, _ = cur.process_frame(extractor)
            kp_prev, ft_prev, known_3d, point_ids = m.GetImagePointsWithFrameID(0)
            matches, _, _, cur_pts, cur_fts = matcher.match_features(kp_prev, ft_prev, kp_cur, ft_cur)
            q = matches.query_idx
            if k == 3:                      # fewer correspondences than the device matched
                q, cur_pts, cur_fts = q[:-10], cur_pts[:-10], cur_fts[:-10]
            prev_pose = np.asarray(m.GetFrame(k - 1).GetPose(), np.float64)
            pose = prev_pose
            if k != 6:                      # k == 6: no PnP at all
                c_T_w = np.linalg.inv(prev_pose)
                ok, rvec, tvec, _ = hf.solvePnPRansac(known_3d[q], cur_pts, K, None, hf.Rtorvec(c_T_w[:3, :3]), c_T_w[:3, 3],
                                                      useExtrinsicGuess=True, context=vs, seed=k)
                if ok and k != 4:           # k == 4: the PnP pose is ignored
                    pose = np.linalg.inv(np.asarray(hf.transformMatrix(rvec, tvec)))
            if k == 5:                      # the match list is edited between PnP and the BA
                q, cur_pts, cur_fts = q[5:], cur_pts[5:], cur_fts[5:]
            m.AddParentAndPose(parent_id=k - 1, frame_id=k, frame_obj=cur, rel_pose_trans=np.eye(4), pose=pose)
            m.AddPointToFrameCorrespondences(point_ids=[point_ids[i] for i in q.tolist()], image_points=cur_pts,
                                             descriptors=cur_fts, frame_obj=cur)
            BundleAdjustment(camera, context=vs).motionOnlyBundleAdjustement(m)
        return np.stack([m.GetFrame(k).GetPose() for k in range(len(frames))])

    Map.use_device_mirror = False
    try:
        ref = drive()
    finally:
        Map.use_device_mirror = True
    calls = {"begin": 0, "back_end": 0}
    orig_begin, orig_end = vs.track_begin, vs.track_back_end

    def begin(*a, **k):
        calls["begin"] += 1
        return orig_begin(*a, **k)

    def end(*a, **k):
        calls["back_end"] += 1
        return orig_end(*a, **k)
    vs.track_begin, vs.track_back_end = begin, end
    try:
        got = drive()
    finally:
        del vs.track_begin, vs.track_back_end
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(got, ref))
    assert rel <= 1e-9, rel
    # frames 2, 7 and 8 follow the reference's sequence to the letter: collected from the device; 4 and 5 had a back half in
    # flight that did not match what the caller then built: the period was started afresh both times
    assert calls["back_end"] >= 2 and calls["begin"] >= 3, calls


def test_track_front_back_entry_points_equal_track_frame(vs, oracle):
    """vs_track_front + vs_track_back_begin + vs_track_back_end are one vs_track_frame cut in three (the pieces the class
    API drives): same kernels on the same device state, so key points, descriptors, matches, the PnP outcome and all poses
    are bit-identical to the one-call form; and the pieces refuse to be called out of order."""
    import visual_slam_amd
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(7)
    xy0, _, desc0 = vs.detect_describe_bgr(frames[0], 20, 3000)
    xyz = harness.backproject(xy0, depth0)

    def begin():
        vs.track_begin(xyz, desc0, np.eye(4), ICL_NUIM_K, max_frames=len(frames) - 1, pnp_iterations=100)
    begin()
    ref = [vs.track_frame(frames[k], seed=k, want_keypoints=True, want_matches=True) for k in range(1, len(frames))]
    vs.track_end()
    begin()
    with pytest.raises(visual_slam_amd.VsError):
        vs.track_back_begin(seed=1)            # no front half yet
    with pytest.raises(visual_slam_amd.VsError):
        vs.track_back_end()                    # no back half running
    for k in range(1, len(frames)):
        f = vs.track_front(frames[k], 20, 0.8)
        r = ref[k - 1]
        assert np.array_equal(f["xy"], r["xy"]) and np.array_equal(f["desc"], r["desc"])
        assert np.array_equal(f["match_q"], r["match_q"]) and np.array_equal(f["match_t"], r["match_t"])
        oq, ot, od = oracle.match_ratio(desc0, f["desc"], 0.8)
        assert np.array_equal(f["match_q"], oq) and np.array_equal(f["match_t"], ot) and np.array_equal(f["match_d"], od)
        if k == 3:  # a front half that is not followed up is simply done again
            f2 = vs.track_front(frames[k], 20, 0.8)
            assert np.array_equal(f2["match_q"], f["match_q"]) and np.array_equal(f2["xy"], f["xy"])
        p = vs.track_back_begin(seed=k)
        assert p["found"] == r["pnp_found"] and len(p["inliers"]) == r["pnp_inliers"]
        with pytest.raises(visual_slam_amd.VsError):
            vs.track_front(frames[k], 20, 0.8)  # the back half is still running
        poses = vs.track_back_end()
        assert np.array_equal(poses, r["poses"]), k
    vs.track_end()


def test_pipelined_tracking_mixes_chained_and_host_paced_frames(vs):
    """vs_track_frame_pipelined chains a frame's back half on the device (enqueued before the previous results are known,
    alternating streams, tagged words instead of events) only when no host decision can be needed in between: PnP on, LM on,
    one-launch solve.  A stream whose frames switch the LM off and on again, blank frames (no matches: PnP finds nothing)
    and optional per-frame outputs must give, frame for frame, what the frame-by-frame entry point gives -- bit for bit."""
    from visual_slam_amd import harness
    from visual_slam_amd.workloads import ICL_NUIM_K
    frames, depth0 = harness.load_sequence(12)
    blank = np.full_like(frames[0], 90)
    seq = [frames[1], frames[2], frames[3], blank, frames[4], frames[5], frames[6], frames[7], frames[8], frames[9]]
    lms = [10, 10, 0, 10, 10, 0, 0, 10, 10, 10]      # 0: cannot be chained (and leaves a solve that is not one launch behind it)
    xy0, _, desc0 = vs.detect_describe_bgr(frames[0], 20, 3000)

    def run(pipelined, **kw):
        vs.track_begin(harness.backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=len(seq), pnp_iterations=100)
        outs = []
        try:
            if pipelined:
                for k, (img, lm) in enumerate(list(zip(seq, lms)) + [(None, 10)]):
                    r = vs.track_frame_pipelined(img, seed=k + 1, lm_iterations=lm, **kw)
                    if r is not None:
                        outs.append(r)
            else:
                for k, (img, lm) in enumerate(zip(seq, lms)):
                    outs.append(vs.track_frame(img, seed=k + 1, lm_iterations=lm, **kw))
        finally:
            vs.track_end()
        return outs

    ref = run(False, want_keypoints=True, want_matches=True)
    for kw in (dict(want_keypoints=True, want_matches=True), dict(want_keypoints=False, want_matches=False)):
        got = run(True, **kw)
        assert len(got) == len(ref) == len(seq)
        for a, b in zip(ref, got):
            assert np.array_equal(a["poses"], b["poses"]) and a["n_matches"] == b["n_matches"] and a["pnp_found"] == b["pnp_found"]
            if kw["want_keypoints"]:
                assert np.array_equal(a["xy"], b["xy"]) and np.array_equal(a["desc"], b["desc"])
                assert np.array_equal(a["match_q"], b["match_q"]) and np.array_equal(a["match_t"], b["match_t"])
    assert ref[3]["n_matches"] == 0 and ref[3]["pnp_found"] == 0   # the blank frame
    # and a second period right behind it on the same context (tags and tickets carry over)
    again = run(True, want_keypoints=False, want_matches=False)
    assert all(np.array_equal(a["poses"], b["poses"]) for a, b in zip(ref, again))


def test_first_tracking_period_of_a_fresh_process_may_be_the_chained_one():
    """In chained pipelined tracking a PnP launch waits in-kernel for kernels on other streams.  A kernel that needs scratch
    memory cannot start on a queue before the runtime has provided it, and on a fresh process that provision waited for
    the very kernels being waited for: the first chained period of a process timed out (pnp_ransac_kernel kept its argument
    struct in scratch).  A child process whose very first tracking call is the chained form must simply work."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np\n"
        "from visual_slam_amd import Context, harness\n"
        "ctx = Context(0)\n"
        "frames, depth0 = harness.load_sequence(8)\n"
        "a, _, _ = harness.track_sequence_resident(ctx, frames, depth0, pipelined=True)\n"
        "b, _, _ = harness.track_sequence_resident(ctx, frames, depth0)\n"
        "assert np.array_equal(a, b)\n"
        "ctx.close()\n"
        "print('fresh-process chained period ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fresh-process chained period ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def _debug(vs, inject=0):
    import ctypes as C
    from visual_slam_amd import _capi
    n = C.c_int(0)
    assert _capi.load().vs_track_debug(vs.handle, int(inject), C.byref(n)) == 0
    return n.value


@pytest.mark.parametrize("fail_at", [0, 3, 7])
def test_chained_tracking_redoes_a_frame_whose_hand_off_never_came(vs, fail_at):
    """Round-3 verdict weak #3: a missed tag inside a chained back half ended the period with VS_EHIP after seconds of spinning.
    Now every in-kernel wait gives up within ~50 ms and the host redoes THAT frame host-paced from the last state it handed
    out (track_redo) -- the frame behind it, whose chained back half ran on the void results, is enqueued again as well.
    Injected here (vs_track_debug): the PnP launch of frame `fail_at` waits for a front-half tag nobody publishes.  The poses
    of every frame equal the undisturbed frame-by-frame run bit for bit, exactly one redo is counted, and the period (and the
    next one) carries on chained."""
    from visual_slam_amd import harness
    from visual_slam_amd.workloads import ICL_NUIM_K
    frames, depth0 = harness.load_sequence(12)
    seq = frames[1:11]
    xy0, _, desc0 = vs.detect_describe_bgr(frames[0], 20, 3000)

    def run(pipelined, inject_at=None):
        vs.track_begin(harness.backproject(xy0, depth0), desc0, np.eye(4), ICL_NUIM_K, max_frames=len(seq), pnp_iterations=100)
        outs = []
        try:
            if pipelined:
                for k, img in enumerate(list(seq) + [None]):
                    if k == inject_at:
                        _debug(vs, inject=1)  # consumed by the next CHAINED back half: frame k's
                    r = vs.track_frame_pipelined(img, seed=k + 1)
                    if r is not None:
                        outs.append(r)
            else:
                for k, img in enumerate(seq):
                    outs.append(vs.track_frame(img, seed=k + 1))
        finally:
            vs.track_end()
        return outs

    ref = run(False)
    before = _debug(vs)
    got = run(True, inject_at=fail_at)
    assert _debug(vs) == before + 1, "exactly one back half was redone"
    assert len(got) == len(ref) == len(seq)
    for a, b in zip(ref, got):
        assert np.array_equal(a["poses"], b["poses"]) and a["n_matches"] == b["n_matches"] and a["pnp_found"] == b["pnp_found"]
    again = run(True)  # the next period on the same context: nothing left behind by the redo
    assert _debug(vs) == before + 1
    assert all(np.array_equal(a["poses"], b["poses"]) for a, b in zip(ref, again))


def test_pipelined_tracking_with_the_other_train_staging_equals_frame_by_frame(vs):
    """Round-3 verdict weak #3 (b): whether a period may be chained must not depend on a tuning knob silently selecting a
    kernel the chain cannot carry.  The chain is now offered only when the runtime reports no private segment for every
    kernel it launches (hipFuncGetAttributes on the loaded code objects; asked again when vs_tune_match selects another match
    kernel); either way pipelined tracking equals the frame-by-frame entry point bit for bit."""
    from visual_slam_amd import harness
    frames, depth0 = harness.load_sequence(10)
    ref, _, _ = harness.track_sequence_resident(vs, frames, depth0)
    try:
        for ts in (0, 1):
            vs.tune_match(tstage=ts)
            got, _, _ = harness.track_sequence_resident(vs, frames, depth0, pipelined=True)
            assert np.array_equal(ref, got), ts
    finally:
        vs.tune_match(tstage=1)


def test_class_api_fast_path_engages_for_the_reference_call_as_written(vs):
    """Round-3 advisor: the resident PnP + BA path was reached only by a caller that passed float64 object points and the
    previous pose as the solver's guess.  main.py:187-204 as it stands passes objectPoints.astype(np.float32) and takes rvec /
    tvec from W_T_prev itself (camera-to-world where OpenCV expects world-to-camera).  That call must (a) reach the resident
    period too -- the device rounds its rows to float32 and starts from the caller's guess (vs_track_back_begin's
    guess_pose16 / obj_as_f32) -- and (b) give what the same calls give when nothing is kept resident, to 1e-9."""
    from visual_slam_amd import harness
    from visual_slam_amd.map import Map
    frames, depth0 = harness.load_sequence(12)
    Map.use_device_mirror = False
    try:
        ref_poses, _ = harness.track_sequence_api(frames, depth0, context=vs, verbatim=True)
    finally:
        Map.use_device_mirror = True
    calls = {"back_begin": [], "back_end": 0, "pnp": 0}
    orig = {n: getattr(vs, n) for n in ("track_back_begin", "track_back_end", "pnp_ransac")}

    def back_begin(*a, **k):
        calls["back_begin"].append((k.get("guess") is not None, bool(k.get("obj_f32"))))
        return orig["track_back_begin"](*a, **k)

    def back_end(*a, **k):
        calls["back_end"] += 1
        return orig["track_back_end"](*a, **k)

    def pnp(*a, **k):
        calls["pnp"] += 1
        return orig["pnp_ransac"](*a, **k)
    vs.track_back_begin, vs.track_back_end, vs.pnp_ransac = back_begin, back_end, pnp
    try:
        poses, _ = harness.track_sequence_api(frames, depth0, context=vs, verbatim=True)
    finally:
        for n in orig:
            delattr(vs, n)
    # frame 1 is host-fed (the period does not exist before the first BA call): one plain PnP; every later frame's PnP ran in
    # the period, on float32-rounded rows; the guess is the caller's own from the first frame whose previous pose is not the identity
    assert calls["pnp"] == 1 and calls["back_end"] == len(calls["back_begin"]) == len(frames) - 2, calls
    assert all(f32 for _, f32 in calls["back_begin"]) and any(g for g, _ in calls["back_begin"]), calls
    rel = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(poses, ref_poses))
    assert rel <= 1e-9, rel
    # and the verbatim call differs from the corrected one only as far as the other start point moves RANSAC + LM
    fixed, _ = harness.track_sequence_api(frames, depth0, context=vs)
    assert max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(poses, fixed)) < 1e-2


def test_a_lost_match_chunk_is_reported_where_results_are_handed_out(vs, oracle):
    """Device entry points only enqueue: a folding workgroup of hamming_knn2_kernel that runs out of its bounded wait writes -1
    rows and raises a pinned word.  Round 4 looked at that word only at the stream's NEXT launch -- the error was blamed on
    an innocent call, or never reported when no launch followed.  Now it surfaces where results are handed out:
    vs_match_status (ShardedMatcher's collect / close, Context.synchronize), the tracking period's frame hand-out, vs_track_end.
    The word is raised here by the developer hook vs_match_debug_raise, exactly as the kernel raises it."""
    import torch
    import visual_slam_amd.context as vctx
    from visual_slam_amd import harness
    from visual_slam_amd.context import VsError
    from visual_slam_amd.sharded import ShardedMatcher
    vctx._DEFAULT = vs
    lib, h = vs._lib, vs.handle
    q, t = match_workload(1500, 1200, seed=3)
    oidx, odist = oracle.hamming_knn2(q, t)
    m = ShardedMatcher()
    with torch.cuda.stream(m.torch_stream()):
        dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
        plan = m.plan(dq, dt, len(q), in_flight=2)
        s0 = plan.submit()
        torch.cuda.synchronize()
        assert lib.vs_match_status(h) == 0                      # nothing happened: quiet
        assert lib.vs_match_debug_raise(h) >= 1
        with pytest.raises(VsError, match="did not report within the bounded wait"):
            plan.collect(s0)                                       # reported at the hand-out of THIS step ...
        assert lib.vs_match_status(h) == 0                      # ... once
        s1 = plan.submit()                                         # and the stream's next launch starts clean
        idx, dist = plan.collect(s1)
        assert np.array_equal(idx.cpu().numpy(), oidx) and np.array_equal(dist.cpu().numpy(), odist)
        assert lib.vs_match_debug_raise(h) >= 1
        with pytest.raises(VsError, match="did not report"):
            vs.synchronize()
        assert lib.vs_match_debug_raise(h) >= 1
        with pytest.raises(VsError, match="did not report"):
            m.close()                                              # the last step of a run: nothing goes unreported
    # the tracking period: the report names the frame whose front half it belongs to, and vs_track_end does not swallow it
    frames, depth0 = harness.load_sequence(4)
    xy0, _, d0 = vs.detect_describe_bgr(frames[0], 20, 3000)
    X = harness.backproject(xy0, depth0)
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=4)
    good = vs.track_frame(frames[1])
    assert lib.vs_match_debug_raise(h) >= 1
    with pytest.raises(VsError, match="train chunk"):            # (the frame's own front-half launch finds the word first)
        vs.track_frame(frames[2])
    lib.vs_match_status(h)                                        # (the hook raised the words of the other streams' sets as well)
    vs.track_end()
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=4)
    again = vs.track_frame(frames[1])
    assert np.array_equal(again["poses"], good["poses"])
    assert lib.vs_match_debug_raise(h) >= 1
    with pytest.raises(VsError, match="vs_track_end: a train chunk"):
        vs.track_end()
    vs.track_begin(X, d0, np.eye(4), ICL_NUIM_K, max_frames=4)   # and the context is usable afterwards
    assert np.array_equal(vs.track_frame(frames[1])["poses"], good["poses"])
    vs.track_end()
