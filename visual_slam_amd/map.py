"""Map with the interface of the reference's src/v2/map.py:6-131 (no cv2 / g2o imports).

Besides the reference's dict-of-objects (`frames`, `points_3d`, unchanged and authoritative) the map keeps a
structure-of-arrays mirror of its observation graph: every mutation that goes through a Map
method appends to flat arrays (point slot, frame id, uv, descriptor; xyz per point slot).  `soa()` hands those arrays
to `BundleAdjustment` and `GetImagePointsWithFrameID`, which then need no per-observation Python loop -- the reference
walks P x F dict lookups per frame there (LocalBA.py:207-214, map.py:28-44).  Every Point reports its mutations to the
maps that hold it (change counters, point.py), so a mirror that is still in sync is recognised in O(1); after any edit
behind the map's back (a direct Point.AddFrame / UpdatePoint, a replaced dict) it is re-verified against the object graph
(observation count, point count, xyz identity) and rebuilt from it if needed -- results never depend on the mirror.

Device residency (`_PeriodMirror`): a local map of the reference's tracking loop -- one fixed key
frame, fixed points, one more free frame per image (main.py:181-214) -- is kept resident on the GPU between the
per-frame `BundleAdjustment.motionOnlyBundleAdjustement(local_map)` calls: the call appends only the new frame's
observations and start pose (vs_track_push_frame) instead of rebuilding and re-uploading the whole period.
"""
import copy
from collections import deque as _deque
from functools import partial as _partial

import numpy as np
from operator import itemgetter as _itemgetter

from .point import Point as _Point

_SET_LOC = _Point._loc.__set__          # the slot descriptor: C-level `p._loc = r`
_LOC_REV_FRAMES = __import__("operator").attrgetter("_loc", "_rev", "_frames")
_drain = _partial(_deque, maxlen=0)      # runs an iterator to its end without keeping anything

try:  # rows handed over one by one (one small array per point) packed into one array in C: a third of np.array(list)'s time
    from ._rows import pack as _pack
except ImportError:  # the C helper is not built: numpy's own conversion
    _pack = None
try:  # what a batch of added points holds (positions, counters, observations), read out of the objects in one C pass
    from ._rows import collect as _collect
except ImportError:
    _collect = None


def _rows_of(seq, dtype=None):
    """A list of equally long 1-D arrays -> one 2-D array (a copy), or None when they do not form one (ragged, mixed dtypes)."""
    n = len(seq)
    first = seq[0]
    if _pack is not None and isinstance(seq, list) and isinstance(first, np.ndarray) and first.ndim == 1 and (
            dtype is None or first.dtype == dtype):
        out = np.empty((n, first.shape[0]), first.dtype)
        if _pack(seq, out):
            return out
    try:
        out = np.array(seq, dtype)
    except (ValueError, TypeError):
        return None
    return out if out.ndim == 2 and out.shape[0] == n else None


_SOA_GENERATION = [0]  # every mirror ever built gets the next number: a cache key that, unlike id(), is never recycled


class _UnknownFrames(dict):
    """frame_objs of a mirror that took observations without being told their Frame objects: never consulted."""


_UNKNOWN_FRAMES = _UnknownFrames()


class _SoA:
    """Flat mirror of the observation graph.  Observation rows are kept in chunks and concatenated lazily."""

    def __init__(self):
        _SOA_GENERATION[0] += 1
        self.gen = _SOA_GENERATION[0]
        self.point_slot = {}      # point id -> slot (row of xyz); slots follow points_3d insertion order
        self.xyz = np.zeros((0, 3))
        self.n_points = 0
        self._chunks = []         # (slot int32[k], frame_id int64[k], uv float64[k,2], desc [k,D] or None)
        self._cat = None
        self.n_obs = 0
        self.rev = 0              # sum of Point._rev when the mirror was last known to be in sync
        self.xyz_refs = []        # the location_3d object mirrored in each xyz row (identity check)
        self.batches = []         # chronological log of add_obs calls: (frame id or None for mixed ids, first row, rows)
        self.fid_rows = {}        # frame id -> number of observation rows carrying it
        self.batch_data = []      # per batch: (slot int32[k], uv [k,2]) as they arrived (no concatenation needed to read one)
        self.frame_objs = {}      # frame id -> the Frame object recorded with its observations (None: several different ones)

    def add_point(self, point_id, location):
        slot = self.n_points
        if slot >= self.xyz.shape[0]:
            grown = np.zeros((max(256, 2 * self.xyz.shape[0]), 3))
            grown[:slot] = self.xyz[:slot]
            self.xyz = grown
        self.xyz[slot] = np.asarray(location, np.float64).reshape(3)
        self.point_slot[point_id] = slot
        self.xyz_refs.append(location)
        self.n_points += 1
        return slot

    def note_frames(self, fids, objs):
        """Remembers which Frame object the observations under a frame id were recorded with."""
        fo = self.frame_objs
        for f, o in zip(fids, objs):
            have = fo.get(f, fo)
            if have is fo:
                fo[f] = o
            elif have is not o:
                fo[f] = None

    def add_obs(self, slots, frame_id, uvs, descs, frame_obj=None, noted=None):
        """frame_id: one id for the whole batch, or one per row.  frame_obj: the Frame object(s) of the batch, likewise.
        noted: {frame id: (Frame object, rows)} when the caller knows the batch's frames and their row counts (Map.AddPoints3D):
        the per-row bookkeeping of Frame objects and row counts is then two dict updates."""
        k = len(slots)
        if k == 0:
            return
        if noted is not None:
            if self.frame_objs is not _UNKNOWN_FRAMES:
                self.note_frames(list(noted), [o for o, _ in noted.values()])
        elif frame_obj is None:
            self.frame_objs = _UNKNOWN_FRAMES  # observations without their Frame object: nothing may be inferred from ids
        elif self.frame_objs is not _UNKNOWN_FRAMES:
            try:
                if isinstance(frame_obj, list):   # one Frame object per row: the distinct (frame id, object) pairs, found in C
                    ids = frame_id if isinstance(frame_id, (list, tuple, np.ndarray)) else [frame_id] * k
                    uniq = dict(zip(zip(ids, map(id, frame_obj)), frame_obj))
                    self.note_frames([f for f, _ in uniq], list(uniq.values()))
                else:                             # one Frame object for the whole batch (and one frame id)
                    self.note_frames((frame_id,), (frame_obj,))
            except TypeError:                     # unhashable frame ids
                self.frame_objs = _UNKNOWN_FRAMES
        uv = _rows_of(uvs) if isinstance(uvs, list) else None  # dtype as given (the reference hands float32 keypoints around)
        if uv is None or uv.shape != (k, 2):
            uv = np.asarray(uvs).reshape(k, 2)
        desc = None
        if isinstance(descs, list):
            desc = _rows_of(descs)
        elif descs is not None:
            try:
                desc = np.asarray(descs)
                if desc.ndim != 2 or desc.shape[0] != k:
                    desc = None
            except (ValueError, TypeError):
                desc = None
        if not isinstance(frame_id, np.ndarray) and not np.isscalar(frame_id) and len(frame_id) == k and (
                frame_id.count(frame_id[0]) == k if isinstance(frame_id, list) else all(f == frame_id[0] for f in frame_id)):
            frame_id = frame_id[0]  # one id repeated (a point entering the map with its first observation)
        if np.ndim(frame_id) == 0:
            fids = np.full(k, frame_id, np.int64)
            self.batches.append((int(frame_id), self.n_obs, k))
            self.fid_rows[int(frame_id)] = self.fid_rows.get(int(frame_id), 0) + k
        else:
            fids = np.asarray(frame_id, np.int64)
            self.batches.append((None, self.n_obs, k))
            if noted is not None:
                for f, (_, rows) in noted.items():
                    self.fid_rows[f] = self.fid_rows.get(f, 0) + rows
            else:
                for f in fids.tolist():
                    self.fid_rows[f] = self.fid_rows.get(f, 0) + 1
        slots = np.asarray(slots, np.int32)
        self._chunks.append((slots, fids, uv, desc))
        self.batch_data.append((slots, uv))
        self._cat = None
        self.n_obs += k

    def arrays(self):
        """(slot, frame_id, uv, desc-or-None) over all observations, chronological order."""
        if self._cat is None:
            if not self._chunks:
                self._cat = (np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros((0, 2)), None)
            else:
                descs = [c[3] for c in self._chunks]
                same = all(d is not None for d in descs) and len({d.shape[1:] + (d.dtype,) for d in descs}) == 1
                self._cat = (np.concatenate([c[0] for c in self._chunks]), np.concatenate([c[1] for c in self._chunks]),
                             np.concatenate([c[2] for c in self._chunks]), np.concatenate(descs) if same else None)
                self._chunks = [self._cat]
        return self._cat


def mirror_of_points(points_3d, cell=None):
    """The flat mirror of a dict of Point objects (point id -> Point with .location_3d and .frames = {frame id: (Frame,
    uv, descriptor)}), built by one walk over the objects.  `cell`: the change counter of the map that will own the mirror."""
    s = _SoA()
    for pid, p in points_3d.items():
        if cell is not None and hasattr(p, "_adopt"):
            p._adopt(cell)
        slot = s.add_point(pid, p.location_3d)
        s.rev += getattr(p, "_rev", 0)
        if p.frames:
            fids = list(p.frames.keys())
            fr = p.frames
            s.add_obs([slot] * len(fids), fids, [fr[f][1] for f in fids], [fr[f][2] for f in fids], [fr[f][0] for f in fids])
    return s


class _LazyPoints(dict):
    """`points_3d` of a local map as Map.GetCopyOfPointObjects hands it out (map.py:60-69: a copy of every point the key frame
    sees, each keeping only that frame's observation) -- with the copies made when somebody LOOKS at them.  The reference's
    tracking loop never does: between two key frames it reads the local map through GetImagePointsWithFrameID and adds to it
    through AddPointToFrameCorrespondences (main.py:183-210), both served by the structure-of-arrays mirror, which is seeded
    from the same arrays (Map.Store3DPoints).  Any dict access other than len() creates the Point objects -- ids, positions
    (own copies), the one (Frame, uv, descriptor) triple each -- and from then on this is an ordinary dict."""
    __slots__ = ("_seed", "_cell")

    def __init__(self, seed):
        dict.__init__(self)
        self._seed = seed     # (ids, xyz [n,3] own copy, its rows as a list, uv [n,2], desc [n,D], frame id, Frame object, revs) or None
        self._cell = None     # change counters of the Map that adopted this dict (Store3DPoints)

    def _make(self):
        seed, self._seed = self._seed, None
        if seed is None:
            return
        from .point import Point
        ids, _, rows, uv, desc, fid, frame_obj, revs = seed
        cells = () if self._cell is None else (self._cell,)
        new = Point.__new__
        put = dict.__setitem__
        for pid, X, u, d, rev in zip(ids, rows, list(uv), list(desc), revs):
            q = new(Point)
            q.ID, q._loc, q._frames, q._rev, q._cells = pid, X, {fid: (frame_obj, u, d)}, rev, cells
            put(self, pid, q)

    def __len__(self):
        return len(self._seed[0]) if self._seed is not None else dict.__len__(self)


def _lazy(name):
    f = getattr(dict, name)

    def method(self, *a, **k):
        if self._seed is not None:
            self._make()
        return f(self, *a, **k)
    method.__name__ = name
    return method


for _n in ("__getitem__", "__setitem__", "__delitem__", "__iter__", "__contains__", "__eq__", "__ne__", "__repr__", "__reversed__",
           "__or__", "__ror__", "__ior__", "keys", "values", "items", "get", "pop", "popitem", "setdefault", "update", "copy", "clear"):
    setattr(_LazyPoints, _n, _lazy(_n))


class _PeriodMirror:
    """One local map's tracking period resident on the GPU (vs_track_begin / vs_track_push_frame), fed incrementally by
    BundleAdjustment.motionOnlyBundleAdjustement.  It mirrors a map of the shape the reference's tracking loop builds
    (main.py:153-214): the first frame is the fixed key frame, every other frame is free, all points are fixed; the
    observations of each free frame arrived as one AddPointToFrameCorrespondences batch.  Anything else -- and any edit
    to what has already been sent (points, earlier observations, earlier poses) -- makes `sync` start the period afresh
    or decline, in which case the caller takes the general path."""

    kInitialFrames = 32

    def __init__(self, ctx, K4):
        self.ctx, self.K4 = ctx, tuple(float(v) for v in K4)
        self.cap = 0
        self.pushed = []       # frame ids sent so far, in order
        self.consumed = 0      # SoA batches consumed so far
        self.written = {}      # frame id -> pose array object this mirror wrote into the Frame
        self.state = None      # (points dict id, n_points, geometry counter, observation counter, key id, key pose value)
        # Front half / PnP of the NEXT frame run inside the resident period when the period holds the key frame's real
        # descriptors (see _begin): `key` = (uv, desc) arrays of the key frame as GetImagePointsWithFrameID hands them out,
        # `spec` = what the device has already worked out for the frame in flight (see speculate_front / speculate_back)
        self.key = None
        self.max_kp = 0
        self.xyz = None
        self.spec = None
        self.unused = 0        # front halves nobody followed up on (then no more are attempted for this map)
        self._bidx = None      # index over the mirror's batch log (see solve)
        self.last_pose = None

    kMaxKeypoints = 3000  # FeatureExtractor's default; an extractor with another cap takes the plain path

    def _begin(self, map_, soa, key_id, key_frame, cap):
        P = soa.n_points
        # The key frame's descriptors, one per map point in point order, make the period able to match a new frame and run
        # PnP-RANSAC on the device (speculate_front / speculate_back).  They are there when every point of the local map
        # is seen by the key frame -- the shape main.py:333-345 builds; otherwise the period only takes host-fed frames.
        desc, key, max_kp, pnp = np.zeros((P, 32), np.uint8), None, 2, 0
        try:
            uv, kd, _, _ = map_.GetImagePointsWithFrameID(key_id)
            if isinstance(kd, np.ndarray) and kd.dtype == np.uint8 and kd.shape == (P, 32) and len(uv) == P:
                desc, key, max_kp, pnp = kd, (uv, kd), self.kMaxKeypoints, 100
        except Exception:
            pass
        self.ctx.track_begin(soa.xyz[:P], desc, np.asarray(key_frame.GetPose(), np.float64), self.K4,
                             max_frames=cap, max_kp=max_kp, pnp_iterations=pnp)
        self.ctx._track_owner = self
        self.cap, self.pushed, self.consumed, self.written = cap, [], 0, {}
        self.key, self.max_kp, self.xyz, self.spec, self.unused = key, max_kp, np.array(soa.xyz[:P], np.float64), None, 0
        self.last_pose = np.array(key_frame.GetPose(), np.float64)
        self.state = (id(map_.points_3d), P, map_._cell[0], map_._cell[1], key_id, np.array(key_frame.GetPose(), np.float64))

    # ---- the frame in flight: FeatureExtractor.compute_features -> speculate_front, FeatureMatcher.match_features ->
    # spec_matches, solvePnPRansac -> speculate_back, motionOnlyBundleAdjustement -> solve() collects.  Every step checks
    # that the caller's arguments are what the device worked on; a caller that does anything else gets the plain path,
    # and a back half that turns out not to match restarts the period from the map (solve()).
    def _alive(self):
        ctx = self.ctx
        return getattr(ctx, "_track_owner", None) is self and ctx._track is not None and self.state is not None

    def speculate_front(self, img, thr, max_kp):
        """-> (xy, desc) of `img` computed inside the resident period, or None (no such period / not applicable)."""
        if self.key is None or self.unused or not self._alive() or max_kp != self.max_kp or len(self.pushed) + 1 > self.cap:
            return None
        if self.spec is not None:
            if self.spec["stage"] == 2:
                return None  # a back half nobody collected: leave it to solve() to sort out
            self.unused += 1  # the previous front half was never followed up: this caller's frames go elsewhere
            self.spec = None
            return None
        r = self.ctx.track_front(img, thr, 0.8)
        self.spec = dict(stage=1, matched=False, ratio=0.8, **r)
        return r["xy"], r["desc"]

    def spec_matches(self, kp1, desc1, kp2, desc2, ratio):
        """The front half's matches if (kp1, desc1) is the key frame and (kp2, desc2) the frame in flight, else None."""
        sp = self.spec
        if sp is None or sp["stage"] != 1 or desc2 is not sp["desc"] or kp2 is not sp["xy"] or ratio != sp["ratio"]:
            return None
        uv, kd = self.key
        if desc1 is not kd and not (getattr(desc1, "shape", None) == kd.shape and np.array_equal(desc1, kd)):
            return None
        sp["matched"] = True
        # what the caller is expected to hand to solvePnPRansac (main.py:189-197): the key frame's 3-D points and this
        # frame's image points of the matches -- gathered once, here
        sp["obj_expect"] = self.xyz[sp["match_q"]]
        sp["img_expect"] = sp["xy"][sp["match_t"]]
        return sp["match_q"], sp["match_t"], sp["match_d"]

    kLmIterations, kHuber = 10, float(np.sqrt(5.991))  # the motion-only BA speculate_back enqueues behind the PnP (LocalBA's constants)

    def speculate_back(self, obj, img, K4, pose0, iterations, reproj_err, confidence, seed):
        """PnP-RANSAC inside the period if the call is the one main.py:196-197 makes on the matches in flight, else None.

        Two spellings of that call are recognised.  The object points may be the map's float64 rows or -- as the reference
        writes it, objectPoints=known_3d_matched[...].astype(np.float32) -- those rows rounded to float32 (the device then
        rounds its resident rows the same way: same values into the same arithmetic as the plain path).  The extrinsic guess
        may be the previous pose, or ANY other pose: main.py:193-194 builds rvec / tvec from W_T_prev itself (camera-to-world
        where OpenCV expects world-to-camera), so what reaches the solver is the inverse of the previous pose; whatever the
        caller passed is what the device starts from."""
        sp = self.spec
        if (sp is None or sp["stage"] != 1 or not sp["matched"] or not self._alive() or iterations != 100
                or tuple(float(v) for v in K4) != self.K4):
            return None
        mq = sp["match_q"]
        if obj.shape != (len(mq), 3) or img.shape != (len(mq), 2) or len(mq) < 5 or not np.array_equal(img, sp["img_expect"]):
            return None
        want = sp["obj_expect"]
        if np.array_equal(obj, want):
            f32 = False
        elif np.array_equal(obj, want.astype(np.float32)):  # (exact: float32 -> float64 is lossless)
            f32 = True
        else:
            return None
        pose0 = np.asarray(pose0, np.float64)
        if pose0.shape != (4, 4) or not np.all(np.isfinite(pose0)):
            return None
        guess = None if float(np.abs(pose0 - self.last_pose).max()) <= 1e-9 else pose0
        r = self.ctx.track_back_begin(seed=seed, reproj_err=reproj_err, confidence=confidence, lm_iterations=self.kLmIterations,
                                      huber_delta=self.kHuber, guess=guess, obj_f32=f32)
        sp["stage"], sp["pnp"], sp["lm"] = 2, r, (self.kLmIterations, self.kHuber)
        return r

    def solve(self, map_, huber_delta, max_iterations):
        """Appends what is new and runs the motion-only BA; returns poses [n_frames,4,4] in map.frames order, or None if
        this map is not a tracking period (the caller then builds the general problem)."""
        ctx = self.ctx
        frames = list(map_.frames.items())
        if len(frames) < 2 or not frames[0][1].IsKeyFrame() or any(f.IsKeyFrame() for _, f in frames[1:]):
            return None
        soa = map_.soa()
        P = soa.n_points
        if P < 1:
            return None
        key_id, key_frame = frames[0]
        ids = [fid for fid, _ in frames[1:]]
        if len(set(ids)) != len(ids) or key_id in ids:
            return None
        # which batches belong to which free frame: exactly one scalar-id batch per frame, none with mixed ids.  The batch
        # log of a mirror only grows, so the index over it is kept between calls and extended by the new entries.
        bc = self._bidx
        if bc is None or bc[0] != soa.gen or bc[1] > len(soa.batches):
            bc = self._bidx = [soa.gen, 0, {}, set()]  # mirror generation, batches indexed, fid -> [(b, first, k)], ids in mixed batches
        if bc[1] < len(soa.batches):
            by_fid, mixed = bc[2], bc[3]
            for b in range(bc[1], len(soa.batches)):
                fid, first, k = soa.batches[b]
                if fid is None:
                    _, bf, _, _ = soa.arrays()
                    mixed.update(np.unique(bf[first:first + k]).tolist())
                else:
                    by_fid.setdefault(fid, []).append((b, first, k))
            bc[1] = len(soa.batches)
        if bc[3] and not bc[3].isdisjoint(ids):
            return None
        rows = {}
        for fid in ids:
            got = bc[2].get(fid)
            if got is not None:
                if len(got) > 1:
                    return None
                rows[fid] = got[0]
        state = (id(map_.points_3d), P, map_._cell[0], map_._cell[1], key_id)
        fresh = (getattr(ctx, "_track_owner", None) is not self or ctx._track is None or self.state is None
                 or self.state[:5] != state or not np.array_equal(self.state[5], np.asarray(key_frame.GetPose(), np.float64))
                 or self.pushed != ids[:len(self.pushed)] or len(ids) > self.cap)
        if not fresh:
            for fid, f in frames[1:1 + len(self.pushed)]:
                w = self.written.get(fid)
                if f.GetPose() is not w and not (w is not None and np.array_equal(np.asarray(f.GetPose()), w)):
                    fresh = True  # an earlier pose was edited after the last solve
                    break
                b = rows.get(fid)
                if b is not None and b[0] >= self.consumed:
                    fresh = True  # observations of an already sent frame arrived later
                    break
        sp, self.spec = self.spec, None
        if sp is not None and sp["stage"] == 2 and getattr(ctx, "_track_owner", None) is self and ctx._track is not None:
            # A back half is running on the device (speculate_back).  It is this call's solve iff exactly one new frame
            # arrived with the very matches and (to rounding: the caller took the pose through rvec / tvec) the PnP pose.
            poses = None
            todo = frames[1 + len(self.pushed):]
            if not fresh and len(todo) == 1:
                fid, f = todo[0]
                b = rows.get(fid)
                if b is not None and b[0] >= self.consumed and b[2] == len(sp["match_q"]):
                    sl, obs = soa.batch_data[b[0]]
                    # (... and the solve the device ran is the one asked for: same iteration cap, same kernel width)
                    if (np.array_equal(sl, sp["match_q"]) and np.array_equal(obs, sp["img_expect"])
                            and sp.get("lm") == (int(max_iterations), float(huber_delta))
                            and float(np.abs(np.asarray(f.GetPose(), np.float64) - sp["pnp"]["pose"]).max()) <= 1e-9):
                        poses = ctx.track_back_end()
                        self.pushed.append(fid)
                        self.consumed = len(soa.batches)
                        self.last_pose = np.array(poses[-1])
            if poses is not None:
                return poses
            fresh = True  # the device solved something else than the caller built: start the period afresh from the map
        if fresh:
            if ctx._track is not None and getattr(ctx, "_track_owner", None) is None:
                return None  # the context's resident period was opened explicitly (Context.track_begin): not ours to end
            if ctx._track is not None:
                ctx.track_end()  # ours, or the previous local map's (a new key frame starts a new local map)
            cap = self.kInitialFrames
            while cap < len(ids) + 8:
                cap *= 2
            self._begin(map_, soa, key_id, key_frame, cap)
        poses = None
        todo = frames[1 + len(self.pushed):]
        if not todo:  # nothing new: re-run the solve on the last frame's state is not expressible -> general path
            return None
        for j, (fid, f) in enumerate(todo):
            b = rows.get(fid)
            if b is None:
                sl, obs = np.zeros(0, np.int32), np.zeros((0, 2))
            else:
                sl, obs = soa.batch_data[b[0]]
                obs = np.asarray(obs, np.float64)
                if sl.size > 1 and not np.all(sl[1:] >= sl[:-1]):  # per-camera order of the general path: point-major
                    order = np.argsort(sl, kind="stable")
                    sl, obs = sl[order], obs[order]
            last = j == len(todo) - 1
            poses = ctx.track_push_frame(sl, obs, np.asarray(f.GetPose(), np.float64),
                                         lm_iterations=max_iterations if last else 0, huber_delta=huber_delta)
            self.pushed.append(fid)
        self.consumed = len(soa.batches)
        self.last_pose = np.array(poses[-1])
        return poses

    def wrote(self, fid, pose):
        self.written[fid] = pose


class Map:
    use_device_mirror = True  # False: motionOnlyBundleAdjustement always builds and uploads the whole problem
    use_lazy_copies = True    # False: GetCopyOfPointObjects always walks the objects and copies them at once

    def __init__(self):
        self.frames = {}
        self.points_3d = {}
        # shared with the points this map holds (point.py): [geometry edits, observation edits, flush callback or None,
        # True once any of this map's points is held by a second map as well (sticky)]
        self._cell = [0, 0, None, False]
        self._pending = []        # observation batches not yet written into the Point objects (see _flush)
        self._added = []          # points added since the mirror was last used (see _absorb_added)
        self._soa = _SoA()
        self._soa_points_obj = self.points_3d  # the dict object the mirror was built from
        self._soa_cell = (0, 0)   # counter values at which the mirror was last known to be in sync
        self._img_cache = {}      # frame id -> (state key, answer of GetImagePointsWithFrameID)
        self._dev = None          # _PeriodMirror: this local map's tracking period resident on the GPU

    # ------------------------------------------------------------------ SoA mirror
    def _soa_rebuild(self):
        self._soa = mirror_of_points(self.points_3d, self._cell)
        self._soa_points_obj = self.points_3d
        self._soa_cell = (self._cell[0], self._cell[1])

    def soa(self):
        """The verified SoA mirror.  In sync and untouched since the last look (the points' change counters, the dict
        object and its length say so): O(1).  Otherwise one O(#points) walk of cheap Python (len() and identity per
        point) decides between refreshing the moved xyz rows and a rebuild from the objects."""
        if self._added:
            self._absorb_added()
        s = self._soa
        c = self._cell
        if (self._soa_points_obj is self.points_3d and s.n_obs >= 0 and s.n_points == len(self.points_3d)
                and self._soa_cell[0] == c[0] and self._soa_cell[1] == c[1]):
            return s
        ok = self._soa_points_obj is self.points_3d and s.n_points == len(self.points_3d)
        if ok:
            n = rev = 0
            refs = s.xyz_refs
            for i, p in enumerate(self.points_3d.values()):
                n += len(p.frames)
                rev += p._rev
                if p.location_3d is not refs[i]:  # UpdatePoint rebinds location_3d: refresh that row
                    s.xyz[i] = np.asarray(p.location_3d, np.float64).reshape(3)
                    refs[i] = p.location_3d
            ok = n == s.n_obs and rev == s.rev  # a direct Point.AddFrame (new or overwriting) changes rev
        if not ok:
            self._soa_rebuild()
            s = self._soa
        self._soa_cell = (c[0], c[1])
        return s

    def resident_motion_ba(self, ctx, K4, huber_delta, max_iterations):
        """motionOnlyBundleAdjustement on the device-resident period of this map (see _PeriodMirror): poses [n,4,4] in
        `frames` order, or None when the map does not have the shape of a tracking period."""
        if not self.use_device_mirror or not hasattr(ctx, "track_push_frame"):
            return None
        d = self._dev
        if d is None or d.ctx is not ctx or d.K4 != tuple(float(v) for v in K4):
            d = self._dev = _PeriodMirror(ctx, K4)
        return d.solve(self, huber_delta, max_iterations)

    # ------------------------------------------------------------------ reference API
    def AddFrame(self, frame_id, frame):
        if frame_id in self.frames.keys():
            raise Exception("Duplicate frame warning")
        self.frames[frame_id] = frame

    def GetPointsVisibleToFrames(self, frame_id_list):
        point_id_list = []
        for point_obj in self.points_3d.values():
            if all(point_obj.IsVisibleTo(frame_id) for frame_id in frame_id_list):
                point_id_list.append(point_obj.GetID())
        return point_id_list

    def GetImagePointsWithFrameID(self, frame_id):
        """(uv [P,2], descriptors [P,D], xyz [P,3], ids [P]) of the points seen by frame_id, in dict order
        (map.py:28-44).  Served from the SoA mirror (one boolean mask) when it holds the descriptors, else by one dict
        lookup per point."""
        s = self.soa()
        try:
            key = (s.gen, s.n_points, self._cell[0], self._cell[1], s.fid_rows.get(frame_id, 0))
            hit = self._img_cache.get(frame_id)
        except TypeError:  # unhashable frame id
            key = hit = None
        if hit is not None and hit[0] == key:
            return hit[1]  # nothing that this answer depends on has changed: the same (read-only) arrays again
        slot, fid, uv, desc = s.arrays()
        if desc is not None and s.n_obs:
            sel = np.nonzero(fid == frame_id)[0]
            if sel.size:
                order = np.argsort(slot[sel], kind="stable")  # dict (slot) order; one observation per (point, frame)
                sel = sel[order]
                sl = slot[sel]
                if np.all(np.diff(sl) > 0):
                    try:
                        ids = np.fromiter(s.point_slot.keys(), dtype=np.int64, count=s.n_points)  # (= points_3d's keys, in order)
                        out = (uv[sel], desc[sel], s.xyz[sl].copy(), ids[sl])
                        if key is not None:
                            # the cached answer is handed to every later caller as the same array objects (the descriptor
                            # array's address is also what keeps its device copy resident): read-only, so that an in-place
                            # edit by one caller raises instead of silently changing what the next caller is given
                            for a in out:
                                a.setflags(write=False)
                            self._img_cache = {frame_id: (key, out)}  # one entry: the key frame of the period
                        return out
                    except (TypeError, ValueError):
                        pass  # non-integer point ids: fall through to the object walk
        image_points, descriptors, locations_3d, point_Ids = [], [], [], []
        for point_obj in self.points_3d.values():
            hit = point_obj.frames.get(frame_id)
            if hit is not None:
                image_points.append(hit[1])
                descriptors.append(hit[2])
                locations_3d.append(point_obj.location_3d)
                point_Ids.append(point_obj.ID)
        return np.array(image_points), np.array(descriptors), np.array(locations_3d), np.array(point_Ids)

    def Get3DPointsWithIDs(self, id_list):
        return np.array([self.points_3d[point_id].location_3d for point_id in id_list]).reshape(-1, 3)

    def GetAll3DPoints(self):
        return np.array([p.location_3d for p in self.points_3d.values()]).reshape(-1, 3)

    def GetCopyOfPointObjects(self, frame_id):
        """Copies of the points visible to frame_id, each keeping only that frame's observation (map.py:60-69).
        The reference deep-copies the whole Point first, which drags every observing Frame (images included) along;
        the copy here is of the Point only -- the (Frame, uv, descriptor) tuple of frame_id is shared, as after the
        reference's SubsetOfFrames the other frames are dropped anyway."""
        lazy = self._lazy_copies(frame_id)
        if lazy is not None:
            return lazy
        points = {}
        for point_key, point_obj in self.points_3d.items():
            if point_obj.IsVisibleTo(frame_id):
                point_copy = copy.copy(point_obj)
                point_copy.location_3d = copy.deepcopy(point_obj.location_3d)
                point_copy.frames = point_obj.SubsetOfFrames(frame_id)
                points[point_key] = point_copy
        return points

    def _lazy_copies(self, frame_id):
        """GetCopyOfPointObjects from the mirror, as a _LazyPoints -- or None when the mirror cannot vouch for the answer: it must
        hold the descriptors, one observation at most per (point, frame), and know the Frame object behind every frame id with
        that object still carrying the id (IsVisibleTo, point.py:37-38, asks the Frame OBJECTS for their ids, not the keys)."""
        if not self.use_lazy_copies:
            return None
        s = self.soa()
        fo = s.frame_objs
        if fo is _UNKNOWN_FRAMES or not s.n_obs:
            return None
        try:
            frame_obj = fo.get(frame_id)
            if frame_obj is None or any(o is None or o.ID != f for f, o in fo.items()):
                return None
        except (TypeError, AttributeError):
            return None
        slot, fid, uv, desc = s.arrays()
        if desc is None:
            return None
        sel = np.nonzero(fid == frame_id)[0]
        if sel.size == 0:
            return None
        sl = slot[sel]
        if sl.size > 1 and not np.all(sl[1:] > sl[:-1]):
            order = np.argsort(sl, kind="stable")
            sel, sl = sel[order], sl[order]
            if not np.all(sl[1:] > sl[:-1]):
                return None  # a point observed twice from one frame: the object walk decides
        pts = list(self.points_3d.values())
        try:
            picked = _itemgetter(*sl.tolist())(pts) if sl.size > 1 else (pts[int(sl[0])],)
            ids = [p.ID for p in picked]
            revs = [p._rev + 1 for p in picked]  # (a copy's `frames` is rebound to the one-entry dict: one more edit, point.py)
            keys = _itemgetter(*sl.tolist())(list(self.points_3d.keys())) if sl.size > 1 else (next(iter(self.points_3d.keys())),)
            if list(keys) != ids:
                return None  # a point stored under another key than its own id: the copies' keys are the map's keys
        except AttributeError:
            return None
        xyz = np.array(s.xyz[sl], np.float64)  # the copies' own positions (map.py:66 deep-copies location_3d)
        return _LazyPoints((ids, xyz, list(xyz), uv[sel], desc[sel], frame_id, frame_obj, revs))

    def GetAllPoses(self):
        return [frame_obj.GetPose() for frame_obj in self.frames.values()]

    def AddPoint3D(self, point_id, point_3d):
        pts = self.points_3d
        if point_id in pts:
            raise Exception("Duplicate point3d warning")
        pts[point_id] = point_3d
        cell = self._cell
        try:
            if not point_3d._cells:          # the usual case: a new point, held by no map yet (Point._adopt, inlined)
                point_3d._cells = (cell,)
            else:
                point_3d._adopt(cell)
        except AttributeError:               # a foreign point object: the mirror is rebuilt from the objects when it is needed
            pass
        cell[0] += 1  # the point set changed: cached answers and the device mirror are stale
        self._added.append((point_id, point_3d))  # the mirror absorbs new points in bulk on its next use

    def AddPoints3D(self, point_ids, locations, observations):
        """NOT in the reference (its callers add points one by one, main.py:130-135,312-318): n new points at once, each with one
        observation per entry of `observations` = [(Frame object, uv [n,2], descriptors [n,D]), ...].  The map ends up exactly as after
            for k in range(n):
                p = Point(location=locations[k], id=point_ids[k])
                for frame, uv, desc in observations: p.AddFrame(frame=frame, uv=uv[k], descriptor=desc[k])
                self.AddPoint3D(point_ids[k], p)
        -- the same Point objects (row k of each array as its position / image point / descriptor object), the same dict order, the
        same change counters, the same rows in the same order in the structure-of-arrays mirror -- but the mirror takes the arrays as
        they are instead of reading them back out of n objects, and the objects are filled slot by slot.  One difference: a duplicate
        id raises before ANY point is added (the loop would have added the points in front of it)."""
        pts = self.points_3d
        ids = list(point_ids)
        n = len(ids)
        locations = np.asarray(locations)
        obs = [(f, np.asarray(uv), np.asarray(d)) for f, uv, d in observations]
        if locations.shape != (n, 3) or any(len(uv) != n or len(d) != n for _, uv, d in obs):
            raise ValueError("AddPoints3D: one location and one observation per frame for every point id")
        if n == 0:
            return
        if len(set(ids)) != n or any(i in pts for i in ids):
            raise Exception("Duplicate point3d warning")
        if self._added:
            self._absorb_added()
        s, cell = self._soa, self._cell
        in_sync = (self._soa_points_obj is pts and s.n_obs >= 0 and s.n_points == len(pts)
                   and self._soa_cell[0] == cell[0] and self._soa_cell[1] == cell[1])
        rows = list(locations)                     # row k: the position object of point k (as `location=pts[k]` binds a row view)
        fids = [f.GetID() for f, _, _ in obs]
        m = len(obs)
        per_frame = [(f, list(uv), list(d)) for f, uv, d in obs]
        new, cells = _Point.__new__, (cell,)
        if m == 1:
            f, uvr, dr = per_frame[0]
            fid = fids[0]
            for pid, X, u, d in zip(ids, rows, uvr, dr):
                q = new(_Point)
                q.ID, q._loc, q._frames, q._rev, q._cells = pid, X, {fid: (f, u, d)}, 1, cells
                pts[pid] = q
        elif m == 2 and fids[0] != fids[1]:   # a new point of a key frame: seen from the previous key frame and from this one
            (f0, uv0, de0), (f1, uv1, de1) = per_frame
            i0, i1 = fids
            for pid, X, u0, d0, u1, d1 in zip(ids, rows, uv0, de0, uv1, de1):
                q = new(_Point)
                q.ID, q._loc, q._frames, q._rev, q._cells = pid, X, {i0: (f0, u0, d0), i1: (f1, u1, d1)}, 2, cells
                pts[pid] = q
        else:
            for k, (pid, X) in enumerate(zip(ids, rows)):
                q = new(_Point)
                fr = {}
                for fid, (f, uvr, dr) in zip(fids, per_frame):   # (a frame id listed twice overwrites, as AddFrame would)
                    fr[fid] = (f, uvr[k], dr[k])
                q.ID, q._loc, q._frames, q._rev, q._cells = pid, X, fr, m, cells
                pts[pid] = q
        cell[0] += n  # (AddPoint3D: one per point)
        if not in_sync or len(set(fids)) != m:
            return    # the mirror is verified against the objects (and rebuilt) on its next use
        slot0 = s.n_points
        if slot0 + n > s.xyz.shape[0]:
            grown = np.zeros((max(256, 2 * s.xyz.shape[0], slot0 + n), 3))
            grown[:slot0] = s.xyz[:slot0]
            s.xyz = grown
        s.xyz[slot0:slot0 + n] = locations
        s.point_slot.update(zip(ids, range(slot0, slot0 + n)))
        s.xyz_refs.extend(rows)
        s.n_points += n
        s.rev += n * m
        if m == 1:
            # (copies: the mirror's rows must not alias the caller's arrays -- the loop's rows are packed into fresh arrays, too)
            s.add_obs(np.arange(slot0, slot0 + n, dtype=np.int32), fids[0], np.array(obs[0][1]), np.array(obs[0][2]), obs[0][0])
        elif m > 1:   # point by point, frame by frame: the order in which the loop above records them
            same_uv = len({(uv.dtype, uv.shape[1:]) for _, uv, _ in obs}) == 1
            same_d = len({(d.dtype, d.shape[1:]) for _, _, d in obs}) == 1
            uv_all = np.stack([uv for _, uv, _ in obs], 1).reshape(n * m, -1) if same_uv else [u for k in range(n) for _, uvr, _ in per_frame for u in (uvr[k],)]
            d_all = np.stack([d for _, _, d in obs], 1).reshape((n * m,) + obs[0][2].shape[1:]) if same_d else [x for k in range(n) for _, _, dr in per_frame for x in (dr[k],)]
            try:
                noted = {int(fid): (f, n) for fid, (f, _, _) in zip(fids, obs)} if all(np.ndim(fid) == 0 for fid in fids) else None
            except (TypeError, ValueError):
                noted = None
            s.add_obs(np.repeat(np.arange(slot0, slot0 + n, dtype=np.int32), m), np.tile(np.asarray(fids, np.int64), n) if noted is not None else fids * n,
                      uv_all, d_all, [f for f, _, _ in obs] * n if noted is None else None, noted=noted)
        self._soa_cell = (cell[0], cell[1])

    def _absorb_added(self):
        """Takes the points added since the last look into the mirror: one pass, one observation batch (the reference's
        callers add a few hundred points one by one, main.py:130-135,312-318)."""
        added, self._added = self._added, []
        s, c = self._soa, self._cell
        if not (self._soa_points_obj is self.points_3d and s.n_obs >= 0 and s.n_points + len(added) == len(self.points_3d)
                and self._soa_cell[0] + len(added) == c[0] and self._soa_cell[1] == c[1]):
            return  # something else happened in between: soa() verifies and rebuilds
        # (Pending observation batches -- AddPointToFrameCorrespondences rows not yet written into the Point objects -- never
        # concern the points absorbed here: a batch only names points the mirror already held when it arrived, and every such
        # call absorbs what was added before it.  The new points' own `_frames` dicts, read below, are complete.)
        n = len(added)
        pids, pts = zip(*added)
        got = _collect(pts) if _collect is not None else None
        if got is not None:
            locs, rev_sum, counts, fids, fobjs, uvs, descs = got
            rows = _rows_of(locs, np.float64)
            if rows is not None and rows.shape == (n, 3):
                slot0 = s.n_points
                if slot0 + n > s.xyz.shape[0]:
                    grown = np.zeros((max(256, 2 * s.xyz.shape[0], slot0 + n), 3))
                    grown[:slot0] = s.xyz[:slot0]
                    s.xyz = grown
                s.xyz[slot0:slot0 + n] = rows
                s.point_slot.update(zip(pids, range(slot0, slot0 + n)))
                s.xyz_refs.extend(locs)
                s.n_points += n
                s.rev += rev_sum
                if len(fids) == n and counts.count(1) == n:
                    slots = range(slot0, slot0 + n)
                else:
                    slots = np.repeat(np.arange(slot0, slot0 + n, dtype=np.int32), counts)
                if len(slots):
                    s.add_obs(slots, fids, uvs, descs, fobjs)
                self._soa_cell = (c[0], c[1])
                return
        if not all(hasattr(p, "_frames") for p in pts):
            return  # foreign point objects: soa() rebuilds the mirror from the objects
        locs, revs, frames = map(list, zip(*map(_LOC_REV_FRAMES, pts)))  # three slots of every point in one C-level pass
        rows = _rows_of(locs, np.float64)  # one conversion for the batch
        if rows is not None and rows.shape != (n, 3):
            rows = None
        fobjs = []
        if rows is None:  # ragged / exotic locations: one by one
            slots, fids, uvs, descs = [], [], [], []
            for pid, p in added:
                slot = s.add_point(pid, p.location_3d)
                s.rev += p._rev
                for f, (fo_, uv, d) in p._frames.items():
                    slots.append(slot)
                    fids.append(f)
                    uvs.append(uv)
                    descs.append(d)
                    fobjs.append(fo_)
        else:
            slot0 = s.n_points
            if slot0 + n > s.xyz.shape[0]:
                grown = np.zeros((max(256, 2 * s.xyz.shape[0], slot0 + n), 3))
                grown[:slot0] = s.xyz[:slot0]
                s.xyz = grown
            s.xyz[slot0:slot0 + n] = rows
            s.point_slot.update(zip(pids, range(slot0, slot0 + n)))
            s.xyz_refs.extend(locs)
            s.n_points += n
            s.rev += sum(revs)
            # `frames`: the observations attached before the points entered the map
            if set(map(len, frames)) == {1}:  # the usual case: one observation each (main.py:130-135,312-318)
                fids, triples = zip(*map(next, map(iter, map(dict.items, frames))))
                fobjs, uvs, descs = map(list, zip(*triples))
                fids = list(fids)
                slots = range(slot0, slot0 + n)
            else:
                slots, fids, uvs, descs = [], [], [], []
                for k, fr in enumerate(frames):
                    for f, (fo_, uv, d) in fr.items():
                        slots.append(slot0 + k)
                        fids.append(f)
                        uvs.append(uv)
                        descs.append(d)
                        fobjs.append(fo_)
        if len(slots):
            s.add_obs(slots, fids, uvs, descs, fobjs)
        self._soa_cell = (c[0], c[1])

    def UpdatePose(self, new_pose, frame_id):
        if frame_id in self.frames.keys():
            self.frames[frame_id].UpdatePose(new_pose)
        else:
            raise Exception("No frame yet added")

    def UpdatePoint3D(self, new_point, point_id):
        if point_id in self.points_3d.keys():
            self.points_3d[point_id].UpdatePoint(new_point)
        else:
            raise Exception("No point yet added")

    def _update_points(self, new_points, keep=None):
        """BA write-back for every point of the map at once (what LocalBA.py:189-190 does with one Map.UpdatePoint3D per point):
        row i of new_points [P,3] becomes the position of the i-th point of points_3d (keep: optional boolean mask of the points
        to touch).  Each Point's location_3d ends up rebound to its own row object, exactly as after P UpdatePoint calls -- but
        the mirror takes the block as ONE array copy and STAYS IN SYNC: the next soa() / GetImagePointsWithFrameID does not walk
        the points to find out what moved (that walk cost as much as the solve on the driver's key frames)."""
        pts = self.points_3d
        P = len(pts)
        new_points = np.asarray(new_points)
        if new_points.shape != (P, 3):
            raise ValueError("_update_points: one row per point of the map")
        s = self.soa()  # verified mirror (O(1) when nothing happened behind the map's back)
        rows = list(new_points)  # P row views, one object per point (as `point_obj.UpdatePoint(new_points[i])` would bind)
        cell = self._cell
        fast = keep is None and s.n_points == P
        if fast:
            try:
                if not cell[3]:
                    # no point of this map is held by another map: rebinding `_loc` is all there is to do, and the slot's own
                    # descriptor does it without a Python-level loop (Point.location_3d's setter minus the counter traffic)
                    _drain(map(_SET_LOC, pts.values(), rows))
                else:
                    for p, r in zip(pts.values(), rows):
                        p._loc = r
                        if len(p._cells) != 1:      # held by another map as well: that map's mirror must hear of it
                            for c in p._cells:
                                if c is not cell:
                                    c[0] += 1
            except (AttributeError, TypeError):     # foreign point objects: the plain way
                fast = False
        if not fast:
            for i, p in enumerate(pts.values()):
                if keep is None or keep[i]:
                    p.UpdatePoint(new_points[i])
            return
        s.xyz[:P] = new_points
        s.xyz_refs[:] = rows
        cell[0] += 1                                # geometry changed: cached answers (GetImagePointsWithFrameID) are stale ...
        self._soa_cell = (cell[0], cell[1])         # ... but the mirror is not

    def GetFrame(self, frame_id):
        return self.frames[frame_id]

    def GetPoint(self, point_id):
        return self.points_3d[point_id]

    def visualize_map(self, viewer):
        """map.py:100-107: feeds poses (and the cloud once) to a viewer object; the Pangolin viewer itself is out of
        scope, any object with update_pose(pose=, cloud=, colour=) works."""
        from .LocalBA import Isometry3d
        colour = np.array([[0], [0], [0]]).T
        for i, pose in enumerate(self.GetAllPoses()):
            iso = Isometry3d(np.asarray(pose)[:3, :3], np.asarray(pose)[:3, 3])
            if i == 0:
                viewer.update_pose(pose=iso, cloud=self.GetAll3DPoints(), colour=colour)
            else:
                viewer.update_pose(pose=iso, colour=colour)

    def Store3DPoints(self, points_dict):
        self._flush()
        self._added = []
        if isinstance(points_dict, _LazyPoints) and points_dict._seed is not None and not self.points_3d and points_dict._cell is None:
            # the copies GetCopyOfPointObjects described but has not made (main.py:345): this map adopts the description -- the
            # mirror is filled from its arrays in one go, the Point objects appear if and when somebody looks at them
            ids, xyz, rows, uv, desc, fid, frame_obj, revs = points_dict._seed
            n = len(ids)
            s = _SoA()
            s.xyz = np.array(xyz, np.float64)
            s.n_points = n
            s.point_slot = dict(zip(ids, range(n)))
            if len(s.point_slot) == n:
                s.xyz_refs = list(rows)  # the very row objects the copies' location_3d will be bound to (_LazyPoints._make)
                s.rev = sum(revs)
                s.add_obs(np.arange(n, dtype=np.int32), fid, uv, desc, frame_obj)
                points_dict._cell = self._cell
                self.points_3d = points_dict
                self._soa, self._soa_points_obj = s, points_dict
                self._cell[0] += 1
                self._soa_cell = (self._cell[0], self._cell[1])
                self._img_cache = {}
                return
        self.points_3d = {**self.points_3d, **points_dict}  # new dict object: the mirror is rebuilt on next use

    def AddParentAndPose(self, parent_id, frame_id, frame_obj, rel_pose_trans, pose):
        frame_obj.AddParent(parent_frame_id=parent_id, transition=rel_pose_trans)
        frame_obj.AddPose(init_pose=pose)
        frame_obj.AddID(frame_id)
        self.AddFrame(frame_id=frame_id, frame=frame_obj)

    def _flush(self):
        """Writes the pending AddPointToFrameCorrespondences batches into the Point objects -- what the reference does at
        once, one Point.AddFrame per match (map.py:120-122).  The mirror took the batch as arrays when it arrived; the
        per-point dict entries are only needed when somebody looks at a Point's `frames`, which is what triggers this."""
        self._cell[2] = None
        pending, self._pending = self._pending, []
        pts = self.points_3d
        for point_ids, image_points, descriptors, frame_obj, fid in pending:
            for point_id, uv, desc in zip(point_ids, image_points, descriptors):
                p = pts.get(point_id)
                if p is not None:  # the point may have been discarded since
                    p._frames[fid] = (frame_obj, uv, desc)

    def AddPointToFrameCorrespondences(self, point_ids, image_points, descriptors, frame_obj):
        """map.py:120-122.  The Point objects are updated one by one as in the reference; the SoA mirror takes the whole
        batch as three array appends (a re-observation of the same (point, frame) invalidates it instead)."""
        fid = frame_obj.GetID()
        pts = self.points_3d
        if not isinstance(point_ids, (list, tuple, np.ndarray)):
            point_ids = list(point_ids)  # the reference's zip() takes any iterable; it is walked more than once here
        if self._added:
            self._absorb_added()
        s = self._soa
        # fast path: the mirror is in sync, this frame has no observations yet and no point repeats inside the batch ->
        # the batch goes to the mirror as arrays now and to the Point objects when one of them is looked at
        c = self._cell
        if (self._soa_points_obj is pts and s.n_obs >= 0 and s.n_points == len(pts) and self._soa_cell[0] == c[0]
                and self._soa_cell[1] == c[1] and hasattr(image_points, "__getitem__") and hasattr(descriptors, "__getitem__")):
            try:
                n = len(point_ids)
                if n > 1:  # one C-level lookup for the whole batch
                    slots = np.array(_itemgetter(*(point_ids.tolist() if isinstance(point_ids, np.ndarray) else point_ids))(s.point_slot), np.int32)
                else:
                    slots = np.fromiter((s.point_slot[pid] for pid in point_ids), dtype=np.int32)
                new_frame = s.fid_rows.get(fid, 0) == 0
            except (KeyError, TypeError):
                slots, n, new_frame = None, 0, False
            if slots is not None and new_frame and n and len(image_points) >= n and len(descriptors) >= n and (
                    n == 1 or bool(np.all(slots[1:] > slots[:-1])) or len(set(slots.tolist())) == n):
                s.add_obs(slots, fid, image_points[:n], descriptors[:n], frame_obj)
                self._pending.append((list(point_ids), image_points, descriptors, frame_obj, fid))
                c[2] = self._flush
                return
        self._flush()
        fresh = True
        n = 0
        for point_id, uv, desc in zip(point_ids, image_points, descriptors):
            fr = pts[point_id].frames
            if fid in fr:
                fresh = False
            fr[fid] = (frame_obj, uv, desc)  # what Point.AddFrame stores; the mirror takes the batch below instead
            n += 1
        s = self._soa
        if fresh and n and self._soa_points_obj is pts:
            try:
                slots = [s.point_slot[pid] for pid in point_ids]
                s.add_obs(slots, fid, image_points[:n] if hasattr(image_points, "__getitem__") else list(image_points),
                          descriptors[:n] if hasattr(descriptors, "__getitem__") else None, frame_obj)
            except KeyError:
                s.n_obs = -1  # forces a rebuild
        elif not fresh:
            s.n_obs = -1

    def DiscardOutlierMapPoints(self, n_visible_frames=3):
        self._flush()
        self._added = []
        self.points_3d = {pid: p for pid, p in self.points_3d.items() if p.GetNVisibleFrames() >= n_visible_frames}
