"""Map point: host-side mirror of the reference's `Point` (src/v2/point.py:4-59).

The reference keeps, per 3-D point, its id, its position and a dict `frames` that maps a frame id to the triple
(Frame object, image point uv, descriptor).  The attribute names (`ID`, `frames`, `location_3d`) and the method names are
part of the drop-in contract -- main.py and LocalBA.py read them directly -- so they are kept; everything else is this
package's own: observations are counted in `_rev` so that `Map` can tell when the object graph was edited without going
through it (its structure-of-arrays mirror is then rebuilt, see map.py).
"""
import numpy as np


class Point:
    __slots__ = ("ID", "frames", "location_3d", "_rev")

    def __init__(self, location, id):
        self.location_3d = location
        self.ID = id
        self.frames = dict()
        self._rev = 0  # number of AddFrame calls so far (new or overwriting observations alike)

    # ---- identity / geometry -------------------------------------------------------------------- point.py:11-12,52-56
    def GetID(self):
        """Id under which the map stores this point."""
        return self.ID

    def Get3dPoint(self):
        """Current position estimate (whatever object was stored last: list, ndarray ...)."""
        return self.location_3d

    def UpdatePoint(self, new_location):
        """Rebinds the position (BA write-back); the Map mirror notices the new object by identity."""
        self.location_3d = new_location

    def GetVectorNorm(self):
        """Distance from the world origin (used for the median-depth normalisation, LocalBA.py:178-183)."""
        return np.linalg.norm(self.location_3d)

    # ---- observations --------------------------------------------------------------------------------- point.py:14-50
    def AddFrame(self, frame, uv, descriptor):
        """Records (or overwrites) the observation of this point in `frame`, keyed by the frame's current id."""
        key = frame.GetID()
        self.frames[key] = (frame, uv, descriptor)
        self._rev += 1

    def GetFrame(self, frame_id):
        """The (Frame, uv, descriptor) triple stored under `frame_id`, or None."""
        return self.frames.get(frame_id, None)

    def SubsetOfFrames(self, frame_id):
        """One-entry dict with the observation of `frame_id` (KeyError if there is none, as in the reference)."""
        triple = self.frames[frame_id]
        return {frame_id: triple}

    def GetImagePoint(self, frame_id):
        """(uv, descriptor) of the observation stored under `frame_id`, or None."""
        triple = self.frames.get(frame_id, None)
        return None if triple is None else (triple[1], triple[2])

    def IsVisibleTo(self, frame_id):
        """True if some observing Frame currently carries the id `frame_id`.  Like the reference this looks at the Frame
        objects' ids, not at the dict keys: a Frame may have been re-numbered (Frame.AddID) after it was recorded."""
        return any(observer.ID == frame_id for observer, _, _ in self.frames.values())

    def GetNVisibleFrames(self):
        """Number of recorded observations."""
        return len(self.frames)
