"""BundleAdjustment with the interface of the reference's src/v2/LocalBA.py:20-229, solved by the HIP kernels of
libvslam_hip.so (vs_ba_solve) instead of g2o.

The reference subclasses g2o.SparseOptimizer and fills it through add_pose / add_point / add_edge /
AddScalingEdge; optimize() runs Levenberg-Marquardt with a Schur complement on the marginalised points.  Here the same
calls fill flat arrays (SoA) that go to the GPU in one piece; get_pose()/get_point() read the optimised estimates
back.  Camera, Isometry3d are the helper classes the reference keeps in main.py:24-51.
"""
import numpy as np


class Camera:
    """reference main.py:24-30"""

    def __init__(self, fx, fy, cx, cy, baseline=1):
        self.fx = fx
        self.fy = fy
        self.cx = cx
        self.cy = cy
        self.baseline = baseline


class Isometry3d(object):
    """3d rigid transform -- reference main.py:32-51 (also stands in for g2o.Isometry3d: get_pose().matrix())."""

    def __init__(self, R, t):
        self.R = R
        self.t = t

    def matrix(self):
        m = np.identity(4)
        m[:3, :3] = self.R
        m[:3, 3] = self.t
        return m

    def inverse(self):
        return Isometry3d(self.R.T, -self.R.T @ self.t)

    def __mul__(self, T1):
        R = self.R @ T1.R
        t = self.R @ T1.t + self.t
        return Isometry3d(R, t)

    def orientation(self):
        return self.R

    def position(self):
        return self.t


class RobustKernelHuber:
    """g2o.RobustKernelHuber(delta) stand-in (reference LocalBA.py:82)."""

    def __init__(self, delta=1.0):
        self.delta = float(delta)


class RobustKernelDCS:
    """g2o.RobustKernelDCS() stand-in (reference LocalBA.py:97,115); g2o's default delta is 1."""

    def __init__(self, delta=1.0):
        self.delta = float(delta)


_HUBER_DEFAULT = RobustKernelHuber(np.sqrt(5.991))  # one shared instance, as the reference's default argument
_DCS_DEFAULT = RobustKernelDCS()


class BundleAdjustment:
    def __init__(self, camera, context=None, solver=None):
        """camera: Camera(fx, fy, cx, cy).  `solver` (tests only) replaces the GPU call with a callable of the same
        signature as Context.ba_solve."""
        self.focal_length = (camera.fx, camera.fy)
        self.principal_point = (camera.cx, camera.cy)
        self.baseline = 0
        self.fx, self.fy = camera.fx, camera.fy
        self.cx, self.cy = camera.cx, camera.cy
        self._ctx = context
        self._solver = solver
        self._pose_ids, self._poses, self._pose_fixed = {}, [], []      # vertex id 2*pose_id   (LocalBA.py:60)
        self._point_ids, self._points, self._point_fixed = {}, [], []   # vertex id 2*point_id+1 (LocalBA.py:70)
        self._obs_pose, self._obs_point, self._obs_uv, self._obs_info = [], [], [], []
        self._scale_parent, self._scale_child, self._scale_meas = [], [], []
        self._huber = None
        self._huber_set = False
        self._dcs = 1.0
        self._info_is_identity = True
        self.dropped_edges = 0
        self.result = None

    # ------------------------------------------------------------------ graph construction
    def add_pose(self, pose_id, pose, fixed=False):
        """pose: 4x4 camera-to-world (LocalBA.py:56-65)."""
        if pose_id in self._pose_ids:
            return  # g2o refuses a second vertex with the same id
        self._pose_ids[pose_id] = len(self._poses)
        self._poses.append(np.asarray(pose, np.float64).reshape(4, 4))
        self._pose_fixed.append(1 if fixed else 0)

    def add_point(self, point_id, point, fixed=False, marginalized=True):
        """LocalBA.py:68-77 (points are always marginalised here, as in the reference's calls)."""
        if point_id in self._point_ids:
            print("WARNING: tried to add already existing point!")
            return
        self._point_ids[point_id] = len(self._points)
        self._points.append(np.asarray(point, np.float64).reshape(3))
        self._point_fixed.append(1 if fixed else 0)

    def add_edge(self, point_id, pose_id, measurement, edge_id, information=None, robust_kernel=_HUBER_DEFAULT):
        """EdgeProjectP2MC between point and camera (LocalBA.py:79-94)."""
        pi, ci = self._point_ids.get(point_id), self._pose_ids.get(pose_id)
        if pi is None or ci is None:
            self.dropped_edges += 1  # g2o's add_edge fails silently when a vertex is missing
            return
        delta = None if robust_kernel is None else float(robust_kernel.delta)
        if self._huber_set and delta != self._huber:
            raise NotImplementedError("all projection edges of one problem must share the robust kernel")
        self._huber, self._huber_set = delta, True
        if information is None:
            info = (1.0, 0.0, 1.0)
        else:
            information = np.asarray(information, np.float64)
            info = (information[0, 0], 0.5 * (information[0, 1] + information[1, 0]), information[1, 1])
            if info != (1.0, 0.0, 1.0):
                self._info_is_identity = False
        self._obs_pose.append(ci)
        self._obs_point.append(pi)
        self._obs_uv.append(np.asarray(measurement, np.float64).reshape(2))
        self._obs_info.append(info)

    def add_edge_between_poses(self, parent_id, child_id, measurement, information=np.eye(6),
                               robust_kernel=_DCS_DEFAULT):
        """EdgeSE3 (LocalBA.py:97-113).  Never called by the reference's live code (only the scaling
        edge is used), not part of the GPU path."""
        raise NotImplementedError("EdgeSE3 pose-pose edges are unused by the reference's hot path and not implemented")

    def AddScalingEdge(self, parent_id, child_id, measurement, information=np.eye(1), robust_kernel=_DCS_DEFAULT):
        """EdgeSBAScale: measurement = |translation of the stored relative transform| (LocalBA.py:115-131)."""
        a, b = self._pose_ids.get(parent_id), self._pose_ids.get(child_id)
        if a is None or b is None:
            self.dropped_edges += 1
            return
        info = float(np.asarray(information).reshape(-1)[0])
        if info != 1.0:
            raise NotImplementedError("scale edges carry information 1 in the reference")
        if robust_kernel is None:
            raise NotImplementedError("scale edges use RobustKernelDCS in the reference")
        self._dcs = float(robust_kernel.delta)
        self._scale_parent.append(a)
        self._scale_child.append(b)
        self._scale_meas.append(float(np.linalg.norm(np.asarray(measurement, np.float64)[:3, 3])))

    # ------------------------------------------------------------------ solve
    def optimize(self, max_iterations=10, verbose=True):
        """initialize_optimization() + optimize(max_iterations) (LocalBA.py:39-42)."""
        if not self._poses:
            self.result = None
            return
        solver = self._solver
        if solver is None:
            from .context import default_context
            solver = (self._ctx or default_context()).ba_solve
        n_obs = len(self._obs_pose)
        points = self._points if isinstance(self._points, np.ndarray) else (
            np.stack(self._points) if self._points else np.zeros((0, 3)))
        uv = self._obs_uv if isinstance(self._obs_uv, np.ndarray) else (
            np.stack(self._obs_uv) if n_obs else np.zeros((0, 2)))
        self.result = solver(
            np.stack(self._poses), np.asarray(self._pose_fixed, np.uint8),
            points, np.asarray(self._point_fixed, np.uint8),
            np.asarray(self._obs_pose, np.int32), np.asarray(self._obs_point, np.int32),
            uv, (self.fx, self.fy, self.cx, self.cy),
            huber_delta=self._huber if self._huber else 0.0, max_iterations=max_iterations,
            scale_edges=(self._scale_parent, self._scale_child, self._scale_meas) if self._scale_parent else None,
            obs_info=None if (self._info_is_identity or not len(self._obs_info)) else np.asarray(self._obs_info, np.float64),
            dcs_phi=self._dcs)

    def save_to_file(self, filename):
        """Text dump in g2o's vocabulary (LocalBA.py:44-45); poses as translation + unit quaternion."""
        from scipy.spatial.transform import Rotation
        poses = self.result["poses"] if self.result else self._poses
        points = self.result["points"] if self.result else self._points
        with open(filename, "w") as f:
            for pid, i in self._pose_ids.items():
                q = Rotation.from_matrix(np.asarray(poses[i])[:3, :3]).as_quat()
                t = np.asarray(poses[i])[:3, 3]
                f.write("VERTEX_CAM %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n" % (
                    2 * pid, t[0], t[1], t[2], q[0], q[1], q[2], q[3], self.fx, self.fy, self.cx, self.cy, 0))
                if self._pose_fixed[i]:
                    f.write("FIX %d\n" % (2 * pid))
            for pid, i in self._point_ids.items():
                x = np.asarray(points[i])
                f.write("VERTEX_XYZ %d %.17g %.17g %.17g\n" % (2 * pid + 1, x[0], x[1], x[2]))
            inv_pose = {v: k for k, v in self._pose_ids.items()}
            inv_point = {v: k for k, v in self._point_ids.items()}
            for c, p, uv, info in zip(self._obs_pose, self._obs_point, self._obs_uv, self._obs_info):
                f.write("EDGE_PROJECT_P2MC %d %d %.17g %.17g %.17g %.17g %.17g\n" % (
                    2 * inv_point[p] + 1, 2 * inv_pose[c], uv[0], uv[1], info[0], info[1], info[2]))

    def get_pose(self, pose_id):
        """The optimised estimate; .matrix() gives the 4x4 the reference reads (LocalBA.py:133-134,186,227)."""
        i = self._pose_ids[pose_id]
        m = self.result["poses"][i] if self.result is not None else self._poses[i]
        return Isometry3d(np.array(m[:3, :3]), np.array(m[:3, 3]))

    def get_point(self, point_id):
        """LocalBA.py:136-139 (the reference also prints every point here; that side effect is dropped)."""
        i = self._point_ids[point_id]
        return np.array(self.result["points"][i] if self.result is not None else self._points[i])

    # ------------------------------------------------------------------ SoA graph construction
    def _graph_from_soa(self, map, frame_fixed, points_fixed, with_scale_edges):
        """Builds the same problem as the reference's P x F double loop (LocalBA.py:164-172 / 207-214) from the map's
        structure-of-arrays mirror: poses in map.frames order, points in map.points_3d order, edges point-major then
        frame order -- identical arrays, no per-observation Python (oracle/ref_graph.py holds the double loop itself and
        the tests that both give the same arrays)."""
        if self._poses or len(self._points) or len(self._obs_pose):
            raise RuntimeError("BundleAdjustment: the graph of this optimizer is already populated (the reference "
                               "creates one optimizer per solve, main.py:213,322)")
        if hasattr(map, "soa"):
            s = map.soa()
        else:  # any object with .frames / .points_3d dicts of Frame / Point objects: mirror it once
            from .map import mirror_of_points
            s = mirror_of_points(map.points_3d)
        frame_ids = list(map.frames.keys())
        for frame_id in frame_ids:
            frame_obj = map.frames[frame_id]
            self.add_pose(pose_id=frame_id, pose=frame_obj.GetPose(), fixed=frame_fixed(frame_id, frame_obj))
            if with_scale_edges and frame_id != 0:
                for parent_ID in frame_obj.GetParentIDs():
                    self.AddScalingEdge(parent_id=parent_ID, child_id=frame_id,
                                        measurement=frame_obj.GetTransitionWithParentID(parent_ID))
        P = s.n_points
        self._point_ids = dict(s.point_slot)  # point id -> row: the keys of map.points_3d in their order (the mirror's own index)
        self._points = np.array(s.xyz[:P], dtype=np.float64)
        self._point_fixed = np.full(P, 1 if points_fixed else 0, np.uint8)
        slot, fid, uv, _ = s.arrays()
        if len(slot):
            ids = np.asarray(frame_ids)
            srt = np.argsort(ids, kind="stable")
            pos = np.searchsorted(ids[srt], fid)
            pos = np.clip(pos, 0, len(ids) - 1)
            known = ids[srt][pos] == fid           # observations from frames that are not in this map are skipped
            fpos = srt[pos]                        # position of the frame in map.frames order = pose index
            sel = np.nonzero(known)[0]
            order = sel[np.lexsort((fpos[sel], slot[sel]))]  # point-major, frames in map order
            self._obs_pose = fpos[order].astype(np.int32)
            self._obs_point = slot[order].astype(np.int32)
            self._obs_uv = np.asarray(uv[order], np.float64)
        else:
            self._obs_pose, self._obs_point, self._obs_uv = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2))
        self._obs_info = []
        self._huber, self._huber_set = float(_HUBER_DEFAULT.delta), True

    # ------------------------------------------------------------------ the two entry points main.py calls
    def localBundleAdjustement(self, map, last_keyframe_id=None, scale=False, BAwindow=5):
        """LocalBA.py:143-190: all frames (frame 0 fixed) + scaling edge per parent, all points, one edge per
        (point, observing frame); optimise; optional normalisation by the median point norm; write back.
        last_keyframe_id (LocalBA.py:147-151, never passed by the reference's callers) restricts the points to those
        visible to every frame of the map."""
        frame_ids = map.frames.keys()
        self._graph_from_soa(map, lambda fid_, f_: fid_ == 0, points_fixed=False, with_scale_edges=True)
        keep = None
        if last_keyframe_id is not None:
            visible = set(map.GetPointsVisibleToFrames(frame_ids))
            keep = np.fromiter((pid in visible for pid in map.points_3d.keys()), dtype=bool, count=len(self._points))
            self._drop_points(~keep)
        self.optimize()
        median_depth = 1
        pts = self.result["points"] if keep is None else self.result["points"][keep]
        if scale:
            median_depth = np.median(np.linalg.norm(pts, axis=1))
        for frame_id in frame_ids:
            new_pose = self.get_pose(frame_id).matrix()
            new_pose[0:3, 3] /= median_depth
            map.UpdatePose(new_pose=new_pose, frame_id=frame_id)
        new_points = self.result["points"] / median_depth
        if hasattr(map, "_update_points"):
            map._update_points(new_points, keep)  # all points at once; the map's mirror stays in sync (map.py)
        else:
            for i, point_obj in enumerate(map.points_3d.values()):
                if keep is None or keep[i]:
                    point_obj.UpdatePoint(new_points[i])  # what map.UpdatePoint3D does, without the per-point dict lookups

    def keyframePoseAdjustement(self, map):
        """NOT IN THE REFERENCE (slam.run_sequence(keyframe_ba="poses_only")): localBundleAdjustement's graph -- all frames, frame 0
        fixed, a scaling edge per parent, one edge per (point, observing frame) -- with every point held FIXED; the poses are written
        back, the points are left alone.  The reference frees every point (LocalBA.py:165), which on real data with centimetre
        baselines lets the map's scale drift and collapse."""
        frame_ids = map.frames.keys()
        self._graph_from_soa(map, lambda fid_, f_: fid_ == 0, points_fixed=True, with_scale_edges=True)
        self.optimize()
        for frame_id in frame_ids:
            map.UpdatePose(new_pose=self.get_pose(frame_id).matrix(), frame_id=frame_id)

    def _drop_points(self, drop):
        """Removes the observations of the masked points from the problem and fixes those points (they then take no
        part in the solve, like vertices that were never added)."""
        if not drop.any():
            return
        sel = ~drop[self._obs_point]
        self._obs_pose, self._obs_point, self._obs_uv = self._obs_pose[sel], self._obs_point[sel], self._obs_uv[sel]
        self._point_fixed = np.where(drop, 1, self._point_fixed).astype(np.uint8)

    def motionOnlyBundleAdjustement(self, map, scale=False, save=False):
        """LocalBA.py:195-229: key frames and all points fixed, every other pose free; write back poses only."""
        frame_ids = map.frames.keys()
        if self._solver is None and not scale and hasattr(map, "resident_motion_ba"):
            # a local map of the tracking loop stays resident on the GPU between these calls (map.py, _PeriodMirror):
            # only the new frame's observations and start pose travel
            from .context import default_context
            poses = map.resident_motion_ba(self._ctx or default_context(), (self.fx, self.fy, self.cx, self.cy),
                                           float(_HUBER_DEFAULT.delta), 10)
            if poses is not None:
                self._pose_ids = {fid: i for i, fid in enumerate(frame_ids)}
                self.result = {"poses": poses, "points": None}
                # every frame gets its own 4x4 (rows of one fresh copy: distinct memory, so an in-place edit of one pose by
                # the caller touches no other); the mirror remembers which objects it wrote to recognise later edits
                views = list(np.array(poses))
                for f, new_pose in zip(map.frames.values(), views):
                    f.UpdatePose(new_pose)
                map._dev.written.update(zip(frame_ids, views))
                return
        self._graph_from_soa(map, lambda fid_, f_: bool(f_.IsKeyFrame()), points_fixed=True, with_scale_edges=False)
        self.optimize()
        median_depth = 1
        if scale:
            median_depth = np.median(np.linalg.norm(self.result["points"], axis=1))
        for frame_id in frame_ids:
            new_pose = self.get_pose(frame_id).matrix()
            new_pose[0:3, 3] /= median_depth
            map.UpdatePose(new_pose=new_pose, frame_id=frame_id)
