"""Map with the interface of the reference's src/v2/map.py:6-131 (no cv2 / g2o imports).

Besides the reference's dict-of-objects (`frames`, `points_3d`, unchanged and authoritative) the map keeps a
structure-of-arrays mirror of its observation graph: every mutation that goes through a Map
method appends to flat arrays (point slot, frame id, uv, descriptor; xyz per point slot).  `soa()` hands those arrays
to `BundleAdjustment` and `GetImagePointsWithFrameID`, which then need no per-observation Python loop -- the reference
walks P x F dict lookups per frame there (LocalBA.py:207-214, map.py:28-44).  The mirror is checked against the object
graph (observation count, point count, xyz identity) on every use and rebuilt from it when anything was changed behind
the map's back (e.g. a direct Point.AddFrame), so results never depend on it.
"""
import copy

import numpy as np


class _SoA:
    """Flat mirror of the observation graph.  Observation rows are kept in chunks and concatenated lazily."""

    def __init__(self):
        self.point_slot = {}      # point id -> slot (row of xyz); slots follow points_3d insertion order
        self.xyz = np.zeros((0, 3))
        self.n_points = 0
        self._chunks = []         # (slot int32[k], frame_id int64[k], uv float64[k,2], desc [k,D] or None)
        self._cat = None
        self.n_obs = 0
        self.rev = 0              # sum of Point._rev when the mirror was last known to be in sync
        self.xyz_refs = []        # the location_3d object mirrored in each xyz row (identity check)

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

    def add_obs(self, slots, frame_id, uvs, descs):
        """frame_id: one id for the whole batch, or one per row."""
        k = len(slots)
        if k == 0:
            return
        uv = np.asarray(uvs).reshape(k, 2)  # dtype as given (the reference hands float32 keypoints around)
        desc = None
        if descs is not None:
            try:
                desc = np.asarray(descs)
                if desc.ndim != 2 or desc.shape[0] != k:
                    desc = None
            except (ValueError, TypeError):
                desc = None
        fids = np.full(k, frame_id, np.int64) if np.ndim(frame_id) == 0 else np.asarray(frame_id, np.int64)
        self._chunks.append((np.asarray(slots, np.int32), fids, uv, desc))
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


class Map:
    def __init__(self):
        self.frames = {}
        self.points_3d = {}
        self._soa = _SoA()
        self._soa_points_obj = self.points_3d  # the dict object the mirror was built from

    # ------------------------------------------------------------------ SoA mirror
    def _soa_rebuild(self):
        s = _SoA()
        for pid, p in self.points_3d.items():
            slot = s.add_point(pid, p.location_3d)
            s.rev += getattr(p, "_rev", 0)
            if p.frames:
                fids = list(p.frames.keys())
                s.add_obs([slot] * len(fids), fids, [p.frames[f][1] for f in fids], [p.frames[f][2] for f in fids])
        self._soa = s
        self._soa_points_obj = self.points_3d

    def soa(self):
        """The verified SoA mirror.  Verification is O(#points) of cheap Python (len() and identity per point); a
        mismatch -- the object graph was edited without going through the Map -- triggers a rebuild from the objects."""
        s = self._soa
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
        return s

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
        slot, fid, uv, desc = s.arrays()
        if desc is not None and s.n_obs:
            sel = np.nonzero(fid == frame_id)[0]
            if sel.size:
                order = np.argsort(slot[sel], kind="stable")  # dict (slot) order; one observation per (point, frame)
                sel = sel[order]
                sl = slot[sel]
                if np.all(np.diff(sl) > 0):
                    try:
                        ids = np.fromiter(self.points_3d.keys(), dtype=np.int64, count=s.n_points)
                        return uv[sel], desc[sel], s.xyz[sl].copy(), ids[sl]
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
        points = {}
        for point_key, point_obj in self.points_3d.items():
            if point_obj.IsVisibleTo(frame_id):
                point_copy = copy.copy(point_obj)
                point_copy.location_3d = copy.deepcopy(point_obj.location_3d)
                point_copy.frames = point_obj.SubsetOfFrames(frame_id)
                points[point_key] = point_copy
        return points

    def GetAllPoses(self):
        return [frame_obj.GetPose() for frame_obj in self.frames.values()]

    def AddPoint3D(self, point_id, point_3d):
        if point_id in self.points_3d.keys():
            raise Exception("Duplicate point3d warning")
        self.points_3d[point_id] = point_3d
        s = self._soa
        if self._soa_points_obj is self.points_3d and s.n_points == len(self.points_3d) - 1:
            slot = s.add_point(point_id, point_3d.location_3d)
            s.rev += getattr(point_3d, "_rev", 0)
            if point_3d.frames:  # observations attached before the point entered the map (main.py:130-135)
                fids = list(point_3d.frames.keys())
                s.add_obs([slot] * len(fids), fids, [point_3d.frames[f][1] for f in fids],
                          [point_3d.frames[f][2] for f in fids])

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
        self.points_3d = {**self.points_3d, **points_dict}  # new dict object: the mirror is rebuilt on next use

    def AddParentAndPose(self, parent_id, frame_id, frame_obj, rel_pose_trans, pose):
        frame_obj.AddParent(parent_frame_id=parent_id, transition=rel_pose_trans)
        frame_obj.AddPose(init_pose=pose)
        frame_obj.AddID(frame_id)
        self.AddFrame(frame_id=frame_id, frame=frame_obj)

    def AddPointToFrameCorrespondences(self, point_ids, image_points, descriptors, frame_obj):
        """map.py:120-122.  The Point objects are updated one by one as in the reference; the SoA mirror takes the whole
        batch as three array appends (a re-observation of the same (point, frame) invalidates it instead)."""
        fid = frame_obj.GetID()
        pts = self.points_3d
        fresh = True
        n = 0
        for point_id, uv, desc in zip(point_ids, image_points, descriptors):
            p = pts[point_id]
            if fid in p.frames:
                fresh = False
            p.frames[fid] = (frame_obj, uv, desc)
            n += 1
        s = self._soa
        if fresh and n and self._soa_points_obj is pts:
            try:
                slots = [s.point_slot[pid] for pid in point_ids]
                s.add_obs(slots, fid, image_points[:n] if hasattr(image_points, "__getitem__") else list(image_points),
                          descriptors[:n] if hasattr(descriptors, "__getitem__") else None)
            except KeyError:
                s.n_obs = -1  # forces a rebuild
        elif not fresh:
            s.n_obs = -1

    def DiscardOutlierMapPoints(self, n_visible_frames=3):
        self.points_3d = {pid: p for pid, p in self.points_3d.items() if p.GetNVisibleFrames() >= n_visible_frames}
