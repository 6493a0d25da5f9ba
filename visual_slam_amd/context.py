"""Context: one libvslam_hip context (one GPU, one stream) with NumPy-friendly wrappers around the C ABI."""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _capi
from ._capi import VsError, c_f32p, c_f64p, c_i32p, c_u8p, ptr


# Contexts still open when the interpreter exits are closed by an atexit handler, i.e. BEFORE the C runtime's exit
# handlers run (HIP's fat-binary unregistration, a profiler's finalisation): streams, events and device memory are
# released while the HIP runtime is fully alive, never from a destructor that may run after it has shut down.
_LIVE = weakref.WeakSet()


def _close_all():
    for ctx in list(_LIVE):
        try:
            ctx.close()
        except Exception:
            pass


atexit.register(_close_all)


class Context:
    """Owns a vs_ctx.  Raises VsError when no MI355X is available -- the product has no CPU path."""

    def __init__(self, device=0):
        self._lib = _capi.load()
        h = C.c_void_p()
        rc = self._lib.vs_create(C.byref(h), int(device))
        if rc != 0:
            raise VsError(rc, self._lib.vs_last_error(None).decode())
        self._h = h.value  # a plain address: the typed void* tags of _capi take ints
        self._track = None
        self._track_owner = None  # a map's _PeriodMirror when the resident period is driven by the class API
        self._pinned = []
        self.device = int(device)
        _LIVE.add(self)

    def close(self):
        if getattr(self, "_h", None):
            for p in self._pinned:  # arrays handed out by pinned_empty() must not be used after close()
                self._lib.vs_host_free(self._h, p)
            self._pinned = []
            self._lib.vs_destroy(self._h)
            self._h = None
            _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise VsError(rc, self._lib.vs_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return self._lib.vs_stream(self._h)

    def aux_stream(self, index):
        """One of the context's auxiliary compute streams (a hipStream_t address; None beyond the last): created with the
        context, so each has a hardware queue of its own."""
        return self._lib.vs_aux_stream(self._h, int(index))

    def synchronize(self):
        self._chk(self._lib.vs_synchronize(self._h))
        self._chk(self._lib.vs_match_status(self._h))  # device-side match launches report a lost train chunk here

    def match_status(self):
        """Raises VsError when a device-side match launch of this context gave up its bounded wait since the last check
        (vs_match_status).  Cheap: reads pinned words."""
        self._chk(self._lib.vs_match_status(self._h))

    # ------------------------------------------------------------------ test / sweep hooks (state of THIS context)
    def tune_match(self, target_blocks=-1, tstage=-1):
        """Matcher knobs (vs_tune_match; negative = leave as it is): fixed number of workgroups per launch (0 = the
        automatic plan), train rows staged through LDS (1, default) or fed from SGPRs (0)."""
        self._chk(self._lib.vs_tune_match(self._h, int(target_blocks), int(tstage)))

    def tune_ba(self, schur_variant=-1, points_per_workgroup=0, max_slabs=0, motion_variant=-1):
        """BA knobs (vs_tune_ba; negative / zero = leave as it is): Schur kernel of single-tile windows (0 automatic,
        1 tile kernel, 2 ba_schur_small + linearise launch; 3 = as 0, and banded windows of several tiles stay on the tile
        kernel and the dense factorisation instead of ba_schur_window / ba_chol_band), its points per workgroup (from 64
        up: the slab size of ba_schur_window) and slab cap; motion-only form (0 one launch where it applies, 1 one launch
        per LM step)."""
        self._chk(self._lib.vs_tune_ba(self._h, int(schur_variant), int(points_per_workgroup), int(max_slabs),
                                       int(motion_variant)))

    def tune_ba_solve(self, packed_mode):
        """Storage of the reduced system in ba_solve_block (vs_tune_ba_solve): 0 automatic (square up to 126 unknowns, packed lower
        triangle from 127 to 198), 1 never packed (beyond 126: the blocked factorisation in HBM), 2 packed wherever it fits."""
        self._chk(self._lib.vs_tune_ba_solve(self._h, int(packed_mode)))

    def tune_ba_structure(self, on_host):
        """Where a large problem's sparsity structure is built: on the device (default) or by the host passes."""
        self._chk(self._lib.vs_tune_ba_structure(self._h, int(bool(on_host))))

    _SCHUR_NAMES = ("ba_schur", "ba_schur_tile", "ba_schur_small", "ba_schur_window", "none (motion-only)")
    _DENSE_NAMES = ("ba_solve_block", "ba_chol_band", "ba_chol_panel", "ba_solve (element-wise)", "none (motion-only)")

    def ba_last_path(self):
        """dict(schur, dense, unknowns, band, tiles, window_cams) of the newest ba_solve of this context (vs_ba_last_path)."""
        out = np.zeros(6, np.intc)
        self._chk(self._lib.vs_ba_last_path(self._h, out.ctypes.data))
        return dict(schur=self._SCHUR_NAMES[out[0]] if 0 <= out[0] < 5 else None,
                    dense=self._DENSE_NAMES[out[1]] if 0 <= out[1] < 5 else None,
                    unknowns=int(out[2]), band=int(out[3]), tiles=int(out[4]), window_cams=int(out[5]))

    def debug_poison_alloc(self, byte):
        """Developer aid: fill every device buffer this context allocates from now on with `byte` (-1: off)."""
        self._chk(self._lib.vs_debug_poison_alloc(self._h, int(byte)))

    def ba_structure_on_device(self):
        """True when the newest ba_solve of this context built its structure on the device."""
        return bool(self._lib.vs_ba_structure_on_device(self._h))

    # ------------------------------------------------------------------ detection / description (A2-A4)
    def gray_mean3(self, bgr):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, c = bgr.shape
        assert c == 3
        out = np.empty((h, w), np.uint8)
        self._chk(self._lib.vs_gray_mean3_u8(self._h, ptr(bgr, c_u8p), w, h, 3 * w, ptr(out, c_u8p)))
        return out

    def fast9_detect(self, gray, thr=20, border=3, max_kp=3000):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        xy = np.zeros((max(max_kp, 1), 2), np.float32)
        sc = np.zeros(max(max_kp, 1), np.uint8)
        n = C.c_int(0)
        self._chk(self._lib.vs_fast9_detect(self._h, ptr(gray, c_u8p), w, h, w, thr, border, max_kp, ptr(xy, c_f32p),
                                            ptr(sc, c_u8p), C.byref(n)))
        return xy[:n.value].copy(), sc[:n.value].copy()

    def brief256(self, gray, xy):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        n = xy.shape[0]
        desc = np.zeros((max(n, 1), 32), np.uint8)
        keep = np.zeros(max(n, 1), np.int32)
        m = C.c_int(0)
        self._chk(self._lib.vs_brief256(self._h, ptr(gray, c_u8p), w, h, w, ptr(xy, c_f32p), n, ptr(desc, c_u8p),
                                        ptr(keep, c_i32p), C.byref(m)))
        return desc[:m.value].copy(), keep[:m.value].copy()

    def detect_describe_bgr(self, bgr, thr=20, max_kp=3000):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, c = bgr.shape
        assert c == 3
        xy = np.empty((max(max_kp, 1), 2), np.float32)  # only the first n rows are written, and only those are handed out
        sc = np.empty(max(max_kp, 1), np.uint8)
        desc = np.empty((max(max_kp, 1), 32), np.uint8)
        n = C.c_int(0)
        self._chk(self._lib.vs_detect_describe_bgr(self._h, ptr(bgr, c_u8p), w, h, 3 * w, thr, max_kp, ptr(xy, c_f32p),
                                                   ptr(sc, c_u8p), ptr(desc, c_u8p), C.byref(n)))
        # views, not copies: the library remembers the device copy of `desc` by its host address + content fingerprint,
        # so handing this very array to match_ratio / hamming_knn2 skips the upload
        return xy[:n.value], sc[:n.value], desc[:n.value]

    def pinned_empty(self, shape, dtype=np.uint8):
        """NumPy array in pinned host memory (vs_host_alloc): frames placed here are DMA-ed without a staging copy."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._chk(self._lib.vs_host_alloc(self._h, nbytes, C.byref(p)))
        self._pinned.append(p.value)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def pin(self, array):
        """Copy of `array` in pinned host memory."""
        out = self.pinned_empty(array.shape, array.dtype)
        out[...] = array
        return out

    # ------------------------------------------------------------------ matching (A5, A6)
    @staticmethod
    def _desc(a):
        a = np.ascontiguousarray(a, np.uint8)
        if a.ndim != 2 or a.shape[1] != 32:
            a = a.reshape(-1, 32)
        return a

    def hamming_knn2(self, q, t):
        q, t = self._desc(q), self._desc(t)
        nq, nt = q.shape[0], t.shape[0]
        idx = np.zeros((max(nq, 1), 2), np.int32)
        dist = np.zeros((max(nq, 1), 2), np.int32)
        self._chk(self._lib.vs_hamming_knn2(self._h, ptr(q, c_u8p), nq, ptr(t, c_u8p), nt, ptr(idx, c_i32p),
                                            ptr(dist, c_i32p)))
        return idx[:nq].copy(), dist[:nq].copy()

    def match_ratio(self, q, t, ratio=0.8):
        q, t = self._desc(q), self._desc(t)
        nq, nt = q.shape[0], t.shape[0]
        mq = np.zeros(max(nq, 1), np.int32)
        mt = np.zeros(max(nq, 1), np.int32)
        md = np.zeros(max(nq, 1), np.int32)
        n = C.c_int(0)
        self._chk(self._lib.vs_match_ratio(self._h, ptr(q, c_u8p), nq, ptr(t, c_u8p), nt, float(ratio), ptr(mq, c_i32p),
                                           ptr(mt, c_i32p), ptr(md, c_i32p), C.byref(n)))
        return mq[:n.value].copy(), mt[:n.value].copy(), md[:n.value].copy()

    def hamming_knn2_dev(self, d_q, nq, d_t, nt, d_idx, d_dist, stream=None):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()); enqueues, does not synchronise."""
        self._chk(self._lib.vs_hamming_knn2_dev(self._h, int(d_q), int(nq), int(d_t), int(nt),
                                                int(d_idx), int(d_dist),
                                                int(stream) if stream else None))

    def hamming_knn2_packed_dev(self, d_q, nq, d_t, nt, d_out, stream=None):
        """d_out: int32[nq][4] = (idx0, idx1, dist0, dist1) per query, 16-byte aligned; enqueues only."""
        self._chk(self._lib.vs_hamming_knn2_packed_dev(self._h, int(d_q), int(nq), int(d_t), int(nt),
                                                       int(d_out), int(stream) if stream else None))

    def hamming_knn2_sharded_dev(self, d_q_shard, nq_shard, d_t, nt, d_gathered, per, rank, world, nccl_comm=None,
                                 compute_stream=None, comm_stream=None, done_event=None, after_stream=None):
        """One rank's step of the query-sharded match (vs_hamming_knn2_sharded_dev): the compute stream waits for
        done_event's previous recording and for after_stream, kernel into this rank's slot of d_gathered
        int32[world*per][4], then (if nccl_comm) one in-place ncclAllGather on comm_stream; enqueues only."""
        def vp(x):
            return int(x) if x else None
        self._chk(self._lib.vs_hamming_knn2_sharded_dev(self._h, int(d_q_shard), int(nq_shard), int(d_t),
                                                        int(nt), int(d_gathered), int(per), int(rank), int(world),
                                                        vp(nccl_comm), vp(compute_stream), vp(comm_stream), vp(done_event),
                                                        vp(after_stream)))

    def match_ratio_dev(self, d_q, nq, d_t, nt, ratio, d_mq, d_mt, d_md, d_n, stream=None):
        self._chk(self._lib.vs_match_ratio_dev(self._h, int(d_q), int(nq), int(d_t), int(nt),
                                               float(ratio), int(d_mq), int(d_mt), int(d_md),
                                               int(d_n), int(stream) if stream else None))

    # ------------------------------------------------------------------ triangulation
    def triangulate_dlt(self, P1, P2, pts1, pts2, T1=None, T2=None):
        """X4 [n,4] (unit, w >= 0) and, when the world-to-camera transforms are given, depth [n,2]."""
        P1 = np.ascontiguousarray(np.asarray(P1, np.float64)[:3, :4])
        P2 = np.ascontiguousarray(np.asarray(P2, np.float64)[:3, :4])
        pts1 = np.ascontiguousarray(pts1, np.float64)
        pts2 = np.ascontiguousarray(pts2, np.float64)
        pts1 = pts1.reshape(-1, pts1.shape[-1]) if pts1.size else pts1.reshape(0, 2)
        pts2 = pts2.reshape(-1, pts2.shape[-1]) if pts2.size else pts2.reshape(0, 2)
        n, stride = pts1.shape[0], max(pts1.shape[1], 2)
        assert pts2.shape == pts1.shape
        X4 = np.zeros((max(n, 1), 4))
        depth = T1p = T2p = None
        if T1 is not None:
            T1 = np.ascontiguousarray(np.asarray(T1, np.float64)[:3, :4])
            T2 = np.ascontiguousarray(np.asarray(T2, np.float64)[:3, :4])
            depth = np.zeros((max(n, 1), 2))
            T1p, T2p = ptr(T1, c_f64p), ptr(T2, c_f64p)
        self._chk(self._lib.vs_triangulate_dlt(self._h, ptr(P1, c_f64p), ptr(P2, c_f64p), ptr(pts1, c_f64p),
                                               ptr(pts2, c_f64p), n, stride, ptr(X4, c_f64p), T1p, T2p,
                                               ptr(depth, c_f64p) if depth is not None else None))
        return (X4[:n], depth[:n]) if depth is not None else X4[:n]

    # ------------------------------------------------------------------ PnP-RANSAC
    def pnp_ransac(self, obj, img, K, pose0, iterations=100, reproj_err=8.0, confidence=0.99, seed=0, refine_iters=10):
        """obj [N,3], img [N,2], K = (fx, fy, cx, cy), pose0 = camera-to-world 4x4 guess.
        Returns dict(found, pose (camera-to-world), inliers)."""
        obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
        img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
        pose0 = np.ascontiguousarray(pose0, np.float64).reshape(16)
        n = obj.shape[0]
        pose = np.zeros(16)
        inl = np.zeros(max(n, 1), np.int32)
        ni, found = C.c_int(0), C.c_int(0)
        fx, fy, cx, cy = (float(v) for v in K)
        self._chk(self._lib.vs_pnp_ransac(self._h, ptr(obj, c_f64p), ptr(img, c_f64p), n, fx, fy, cx, cy,
                                          ptr(pose0, c_f64p), int(iterations), float(reproj_err), float(confidence),
                                          int(seed), int(refine_iters), ptr(pose, c_f64p), ptr(inl, c_i32p),
                                          C.byref(ni), C.byref(found)))
        return dict(found=bool(found.value), pose=pose.reshape(4, 4), inliers=inl[:ni.value].copy())

    # ------------------------------------------------------------------ tracking period on the device
    def track_begin(self, xyz, desc, key_pose, K, max_frames=64, max_kp=3000, pnp_iterations=100):
        """Upload the last key frame's map points (xyz [P,3], desc uint8 [P,32]) and pose; starts a tracking period."""
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        if xyz.shape[0] != desc.shape[0]:
            raise ValueError("track_begin: xyz and desc differ in length")
        key_pose = np.ascontiguousarray(key_pose, np.float64).reshape(16)
        fx, fy, cx, cy = (float(v) for v in K)
        self._chk(self._lib.vs_track_begin(self._h, ptr(xyz, c_f64p), ptr(desc, c_u8p), xyz.shape[0], ptr(key_pose, c_f64p),
                                           fx, fy, cx, cy, int(max_frames), int(max_kp), int(pnp_iterations)))
        self._track_owner = None
        # per-frame result buffers in PINNED memory (the library DMAs into them; a pageable destination is staged), kept by the
        # context across periods and grown when a period needs more
        b = getattr(self, "_track_bufs", None)
        P = xyz.shape[0]
        if b is None or b["max_kp"] < int(max_kp) or b["P"] < P:
            kp_cap, p_cap = max(int(max_kp), b["max_kp"] if b else 0), max(2 * P, b["P"] if b else 0, 1024)
            b = self._track_bufs = dict(max_kp=kp_cap, P=p_cap, xy=self.pinned_empty((kp_cap, 2), np.float32),
                                        desc=self.pinned_empty((kp_cap, 32), np.uint8), mq=self.pinned_empty((p_cap,), np.int32),
                                        mt=self.pinned_empty((p_cap,), np.int32))
        self._track = dict(P=P, max_frames=int(max_frames), max_kp=int(max_kp), poses=np.zeros((int(max_frames) + 1, 16)),
                           xy=b["xy"], desc=b["desc"], mq=b["mq"], mt=b["mt"])

    def track_frame(self, bgr, thr=20, ratio=0.8, reproj_err=8.0, confidence=0.99, seed=0, lm_iterations=10,
                    huber_delta=float(np.sqrt(5.991)), want_keypoints=False, want_matches=True):
        """One frame of the period -> dict(poses [n+1,4,4] (pose 0 = key frame), n_matches, pnp_found, match_q, match_t
        [, xy, desc])."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_frame: no tracking period (call track_begin)")
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        npo, nm, found, nk = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self._lib.vs_track_frame(
            self._h, ptr(bgr, c_u8p), w, h, 3 * w, int(thr), float(ratio), float(reproj_err), float(confidence), int(seed),
            int(lm_iterations), float(huber_delta), ptr(t["poses"], c_f64p), C.byref(npo), C.byref(nm), C.byref(found),
            ptr(t["xy"], c_f32p) if want_keypoints else None, ptr(t["desc"], c_u8p) if want_keypoints else None,
            C.byref(nk), ptr(t["mq"], c_i32p) if want_matches else None, ptr(t["mt"], c_i32p) if want_matches else None))
        out = dict(poses=t["poses"][:npo.value].reshape(-1, 4, 4).copy(), n_matches=nm.value, pnp_found=bool(found.value),
                   pnp_inliers=found.value, n_keypoints=nk.value)
        if want_matches:
            out["match_q"], out["match_t"] = t["mq"][:nm.value].copy(), t["mt"][:nm.value].copy()
        if want_keypoints:
            out["xy"], out["desc"] = t["xy"][:nk.value].copy(), t["desc"][:nk.value].copy()
        return out

    def track_last_frame(self, want_keypoints=True, want_matches=True):
        """The per-frame arrays of the newest frame handed out by track_frame / track_frame_pipelined, fetched afterwards
        (vs_track_last_frame): dict(n_keypoints, n_matches [, xy, desc] [, match_q, match_t]).  For callers that need them only
        for the rare frame that becomes a key frame and pass want_keypoints=False / want_matches=False per frame."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_last_frame: no tracking period (call track_begin)")
        nk, nm = C.c_int(0), C.c_int(0)
        self._chk(self._lib.vs_track_last_frame(
            self._h, ptr(t["xy"], c_f32p) if want_keypoints else None, ptr(t["desc"], c_u8p) if want_keypoints else None,
            C.byref(nk), ptr(t["mq"], c_i32p) if want_matches else None, ptr(t["mt"], c_i32p) if want_matches else None, C.byref(nm)))
        out = dict(n_keypoints=nk.value, n_matches=nm.value)
        if want_matches:
            out["match_q"], out["match_t"] = t["mq"][:nm.value].copy(), t["mt"][:nm.value].copy()
        if want_keypoints:
            out["xy"], out["desc"] = t["xy"][:nk.value].copy(), t["desc"][:nk.value].copy()
        return out

    def track_frame_pipelined(self, bgr, thr=20, ratio=0.8, reproj_err=8.0, confidence=0.99, seed=0, lm_iterations=10,
                              huber_delta=float(np.sqrt(5.991)), want_keypoints=False, want_matches=True):
        """Submit `bgr` (None: flush) and return the result of the PREVIOUS submitted frame (None on the first call): that
        frame's PnP + BA run on the GPU while this frame is uploaded, detected and matched."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_frame_pipelined: no tracking period (call track_begin)")
        if bgr is not None:
            bgr = np.ascontiguousarray(bgr, np.uint8)
            h, w, _ = bgr.shape
            t["keep"] = bgr  # the upload may still be in flight when this call returns
        else:
            h = w = 0
        has, npo, nm, found, nk = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self._lib.vs_track_frame_pipelined(
            self._h, ptr(bgr, c_u8p) if bgr is not None else None, w, h, 3 * w, int(thr), float(ratio), float(reproj_err),
            float(confidence), int(seed), int(lm_iterations), float(huber_delta), C.byref(has), ptr(t["poses"], c_f64p),
            C.byref(npo), C.byref(nm), C.byref(found), ptr(t["xy"], c_f32p) if want_keypoints else None,
            ptr(t["desc"], c_u8p) if want_keypoints else None, C.byref(nk), ptr(t["mq"], c_i32p) if want_matches else None,
            ptr(t["mt"], c_i32p) if want_matches else None))
        if not has.value:
            return None
        out = dict(poses=t["poses"][:npo.value].reshape(-1, 4, 4).copy(), n_matches=nm.value, pnp_found=bool(found.value),
                   pnp_inliers=found.value, n_keypoints=nk.value)
        if want_matches:
            out["match_q"], out["match_t"] = t["mq"][:nm.value].copy(), t["mt"][:nm.value].copy()
        if want_keypoints:
            out["xy"], out["desc"] = t["xy"][:nk.value].copy(), t["desc"][:nk.value].copy()
        return out

    def track_push_frame(self, point_idx, uv, pose, lm_iterations=10, huber_delta=float(np.sqrt(5.991))):
        """Host-fed frame of the period (vs_track_push_frame): observations (map point index, uv) + start pose -> poses
        [n+1,4,4] after the motion-only BA over the whole period."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_push_frame: no tracking period (call track_begin)")
        point_idx = np.ascontiguousarray(point_idx, np.int32)
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        pose = np.ascontiguousarray(pose, np.float64).reshape(16)
        npo = C.c_int(0)
        self._chk(self._lib.vs_track_push_frame(self._h, ptr(point_idx, c_i32p), ptr(uv, c_f64p), point_idx.shape[0],
                                                ptr(pose, c_f64p), int(lm_iterations), float(huber_delta),
                                                ptr(t["poses"], c_f64p), C.byref(npo)))
        return t["poses"][:npo.value].reshape(-1, 4, 4).copy()

    # ---- the resident period fed by the class API: one vs_track_frame cut where main.py:181-214 needs values on the host
    def track_front(self, bgr, thr=20, ratio=0.8):
        """Front half (vs_track_front): detect + describe `bgr`, match the period's map points against it, append the
        matches.  -> dict(xy [n,2] float32, desc [n,32] uint8, match_q, match_t, match_d [M] int32) -- fresh arrays."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_front: no tracking period (call track_begin)")
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        kp, P = t["max_kp"], t["P"]
        xy, desc = np.empty((kp, 2), np.float32), np.empty((kp, 32), np.uint8)
        m = np.empty((3, max(P, 1)), np.int32)
        nk, nm = C.c_int(0), C.c_int(0)
        self._chk(self._lib.vs_track_front(self._h, ptr(bgr, c_u8p), w, h, 3 * w, int(thr), float(ratio), ptr(xy, c_f32p),
                                           ptr(desc, c_u8p), C.byref(nk), m[0].ctypes.data, m[1].ctypes.data,
                                           m[2].ctypes.data, C.byref(nm)))
        n, M = nk.value, nm.value
        return dict(xy=xy[:n], desc=desc[:n], match_q=m[0, :M], match_t=m[1, :M], match_d=m[2, :M])

    def track_back_begin(self, seed=0, reproj_err=8.0, confidence=0.99, lm_iterations=10,
                         huber_delta=float(np.sqrt(5.991)), guess=None, obj_f32=False):
        """PnP-RANSAC on the front half's matches + (enqueued behind it) the motion-only BA; returns the PnP outcome:
        dict(found, pose [4,4] camera-to-world, inliers [m] int32 indices into the match list).  guess: the extrinsic guess as
        a camera-to-world 4x4 (None: the period's previous pose); obj_f32: the object points are rounded to float32 first
        (the reference passes objectPoints.astype(np.float32), main.py:196)."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_back_begin: no tracking period (call track_begin)")
        pose = np.empty(16)
        inl = np.empty(max(t["P"], 1), np.int32)
        found, ni = C.c_int(0), C.c_int(0)
        g = None if guess is None else np.ascontiguousarray(guess, np.float64).reshape(16)
        self._chk(self._lib.vs_track_back_begin(self._h, float(reproj_err), float(confidence), int(seed), int(lm_iterations),
                                                float(huber_delta), None if g is None else ptr(g, c_f64p), 1 if obj_f32 else 0,
                                                C.byref(found), ptr(pose, c_f64p), ptr(inl, c_i32p), C.byref(ni)))
        return dict(found=bool(found.value), pose=pose.reshape(4, 4), inliers=inl[:ni.value])

    def track_back_end(self):
        """Waits for the BA behind track_back_begin -> poses [n+1,4,4] of the period (pose 0 = key frame)."""
        t = self._track
        if t is None:
            raise VsError(-1, "track_back_end: no tracking period (call track_begin)")
        npo = C.c_int(0)
        self._chk(self._lib.vs_track_back_end(self._h, ptr(t["poses"], c_f64p), C.byref(npo)))
        return t["poses"][:npo.value].reshape(-1, 4, 4).copy()

    def track_end(self):
        rc = self._lib.vs_track_end(self._h)   # the period is closed whatever it reports (a lost match chunk of its last frame)
        self._track = None
        self._track_owner = None
        self._chk(rc)

    # ------------------------------------------------------------------ two-view initialisation
    def essential_ransac(self, x1, x2, threshold, prob=0.999, max_iters=1000, seed=0):
        """x1, x2 K-normalised [N,2] -> dict(found, E [3,3] with x2^T E x1 = 0, mask uint8[N] (0/1), n_inliers)."""
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
        n = x1.shape[0]
        if x2.shape[0] != n:
            raise ValueError("essential_ransac: x1 and x2 differ in length")
        E = np.zeros(9)
        mask = np.zeros(max(n, 1), np.uint8)
        ni, found = C.c_int(0), C.c_int(0)
        self._chk(self._lib.vs_essential_ransac(self._h, ptr(x1, c_f64p), ptr(x2, c_f64p), n, float(threshold), float(prob),
                                                int(max_iters), int(seed), ptr(E, c_f64p), ptr(mask, c_u8p), C.byref(ni),
                                                C.byref(found)))
        return dict(found=bool(found.value), E=E.reshape(3, 3), mask=mask[:n].copy(), n_inliers=ni.value)

    def recover_pose(self, E, x1, x2, dist_thresh=50.0):
        """-> dict(R [3,3], t [3], mask uint8[N] (255/0), X [N,4] homogeneous, n_good)."""
        E = np.ascontiguousarray(E, np.float64).reshape(9)
        x1 = np.ascontiguousarray(x1, np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, np.float64).reshape(-1, 2)
        n = x1.shape[0]
        if x2.shape[0] != n:
            raise ValueError("recover_pose: x1 and x2 differ in length")
        R, t = np.zeros(9), np.zeros(3)
        mask = np.zeros(max(n, 1), np.uint8)
        X = np.zeros((max(n, 1), 4))
        ng = C.c_int(0)
        self._chk(self._lib.vs_recover_pose(self._h, ptr(E, c_f64p), ptr(x1, c_f64p), ptr(x2, c_f64p), n, float(dist_thresh),
                                            ptr(R, c_f64p), ptr(t, c_f64p), ptr(mask, c_u8p), ptr(X, c_f64p), C.byref(ng)))
        return dict(R=R.reshape(3, 3), t=t, mask=mask[:n].copy(), X=X[:n].copy(), n_good=ng.value)

    # ------------------------------------------------------------------ bundle adjustment (A9-A16)
    def ba_solve(self, poses, pose_fixed, points, point_fixed, obs_pose, obs_point, obs_uv, K,
                 huber_delta=float(np.sqrt(5.991)), max_iterations=10, scale_edges=None, obs_info=None, dcs_phi=1.0,
                 trial_trace=False):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 16)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        pose_fixed = np.ascontiguousarray(pose_fixed, np.uint8)
        point_fixed = np.ascontiguousarray(point_fixed, np.uint8)
        obs_pose = np.ascontiguousarray(obs_pose, np.int32)
        obs_point = np.ascontiguousarray(obs_point, np.int32)
        obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
        p = _capi.BAProblem()
        p.n_poses, p.n_points, p.n_obs = poses.shape[0], points.shape[0], obs_pose.shape[0]
        p.poses, p.pose_fixed = ptr(poses, c_f64p), ptr(pose_fixed, c_u8p)
        p.points, p.point_fixed = ptr(points, c_f64p), ptr(point_fixed, c_u8p)
        p.obs_pose, p.obs_point, p.obs_uv = ptr(obs_pose, c_i32p), ptr(obs_point, c_i32p), ptr(obs_uv, c_f64p)
        keep = []
        if obs_info is not None:
            obs_info = np.ascontiguousarray(obs_info, np.float64).reshape(-1, 3)
            p.obs_info = ptr(obs_info, c_f64p)
        if scale_edges is not None and len(scale_edges[0]):
            sp = np.ascontiguousarray(scale_edges[0], np.int32)
            sc = np.ascontiguousarray(scale_edges[1], np.int32)
            sm = np.ascontiguousarray(scale_edges[2], np.float64)
            keep = [sp, sc, sm]
            p.n_scale = sp.shape[0]
            p.scale_parent, p.scale_child, p.scale_meas = ptr(sp, c_i32p), ptr(sc, c_i32p), ptr(sm, c_f64p)
        p.fx, p.fy, p.cx, p.cy = (float(v) for v in K)
        p.huber_delta = float(huber_delta) if huber_delta else 0.0
        p.dcs_phi = float(dcs_phi)
        p.max_iterations = int(max_iterations)
        r = _capi.BAResult()
        poses_out = np.zeros_like(poses)
        points_out = np.zeros_like(points)
        chi = np.full(max(max_iterations, 1), np.nan)
        lam = np.full(max(max_iterations, 1), np.nan)
        r.poses_out, r.points_out = ptr(poses_out, c_f64p), ptr(points_out, c_f64p)
        r.chi2_trace, r.lambda_trace = ptr(chi, c_f64p), ptr(lam, c_f64p)
        tt = None
        if trial_trace:  # per-trial rows (lambda, trial chi2, rho, solve ok): test / diagnostic aid
            tt = np.full((max(10 * max_iterations, 1), 4), np.nan)
            r.trial_trace, r.trial_trace_cap = ptr(tt, c_f64p), tt.shape[0]
        self._chk(self._lib.vs_ba_solve(self._h, C.byref(p), C.byref(r)))
        del keep
        out = dict(poses=poses_out.reshape(-1, 4, 4), points=points_out, chi2_trace=chi[:r.iterations].copy(),
                   lambda_trace=lam[:r.iterations].copy(), chi2_initial=r.chi2_initial, chi2_final=r.chi2_final,
                   lambda_final=r.lambda_final, iterations=r.iterations, trials=r.trials, not_pd=r.not_pd,
                   terminated=r.terminated)
        if tt is not None:
            out["trial_trace"] = tt[:min(r.trials, tt.shape[0])].copy()
        return out

    def debug_cholesky(self, S, b):
        """Test hook (vs_ba_debug_cholesky): the device's dense solver of the reduced camera system on its own.
        Returns (ok, x)."""
        S = np.ascontiguousarray(S, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        n = S.shape[0]
        x = np.zeros(n)
        ok = C.c_int(0)
        self._chk(self._lib.vs_ba_debug_cholesky(self._h, ptr(S, c_f64p), n, ptr(b, c_f64p), ptr(x, c_f64p), C.byref(ok)))
        return bool(ok.value), x


_DEFAULT = None


def default_context():
    """Process-wide context on the GPU of this rank (LOCAL_RANK, else device 0)."""
    global _DEFAULT
    if _DEFAULT is None:
        import os
        _DEFAULT = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _DEFAULT
