"""Map with the interface of the reference's src/v2/map.py:6-131 (no cv2 / g2o imports)."""
import copy

import numpy as np


class Map:
    def __init__(self):
        self.frames = {}
        self.points_3d = {}

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
        (map.py:28-44) -- one dict lookup per point instead of the reference's four."""
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
        self.points_3d = {**self.points_3d, **points_dict}

    def AddParentAndPose(self, parent_id, frame_id, frame_obj, rel_pose_trans, pose):
        frame_obj.AddParent(parent_frame_id=parent_id, transition=rel_pose_trans)
        frame_obj.AddPose(init_pose=pose)
        frame_obj.AddID(frame_id)
        self.AddFrame(frame_id=frame_id, frame=frame_obj)

    def AddPointToFrameCorrespondences(self, point_ids, image_points, descriptors, frame_obj):
        for point_id, uv, desc in zip(point_ids, image_points, descriptors):
            self.GetPoint(point_id).AddFrame(frame_obj, uv, desc)

    def DiscardOutlierMapPoints(self, n_visible_frames=3):
        self.points_3d = {pid: p for pid, p in self.points_3d.items() if p.GetNVisibleFrames() >= n_visible_frames}
