"""Point with the interface of the reference's src/v2/point.py:4-59."""
import numpy as np


class Point:
    def __init__(self, location, id):
        self.ID = id
        self.frames = {}  # frame_id -> (Frame, uv, descriptor)   (point.py:8-9)
        self.location_3d = location
        self._rev = 0     # bumped by AddFrame: lets Map notice observations edited behind its back

    def GetID(self):
        return self.ID

    def GetFrame(self, frame_id):
        return self.frames.get(frame_id)

    def SubsetOfFrames(self, frame_id):
        return {frame_id: self.frames[frame_id]}

    def AddFrame(self, frame, uv, descriptor):
        self.frames[frame.GetID()] = (frame, uv, descriptor)
        self._rev += 1

    def UpdatePoint(self, new_location):
        self.location_3d = new_location

    def IsVisibleTo(self, frame_id):
        # the reference scans the values and compares frame.ID (point.py:33-37); the dict is keyed by that same id at
        # insertion time, but a Frame's ID may be re-assigned later (Frame.AddID), so keep the scan semantics
        for frame, uv, descriptor in self.frames.values():
            if frame_id == frame.ID:
                return True
        return False

    def GetImagePoint(self, frame_id):
        ret = self.frames.get(frame_id)
        if ret is not None:
            _, uv, descriptor = ret
            return (uv, descriptor)
        return None

    def Get3dPoint(self):
        return self.location_3d

    def GetVectorNorm(self):
        return np.linalg.norm(self.location_3d)

    def GetNVisibleFrames(self):
        return len(self.frames)
