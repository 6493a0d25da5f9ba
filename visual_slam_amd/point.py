"""Map point: host-side mirror of the reference's `Point` (src/v2/point.py:4-59).

The reference keeps, per 3-D point, its id, its position and a dict `frames` that maps a frame id to the triple
(Frame object, image point uv, descriptor).  The attribute names (`ID`, `frames`, `location_3d`) and the method names are
part of the drop-in contract -- main.py and LocalBA.py read them directly -- so they are kept; everything else is this
package's own: every mutation (AddFrame, UpdatePoint, rebinding `frames` / `location_3d`) bumps the change counters of the
maps that hold the point, so that a `Map` can tell in O(1) whether its structure-of-arrays mirror is still in sync with
the object graph (map.py).  Only an in-place edit of the `frames` dict itself would go unseen; no caller does that.
"""
import numpy as np


class Point:
    __slots__ = ("ID", "_frames", "_loc", "_rev", "_cells")

    def __init__(self, location, id):
        self._cells = ()      # cells of the maps that hold this point: [geometry edits, observation edits, flush or None]
        self._loc = location
        self.ID = id
        self._frames = dict()
        self._rev = 0  # number of AddFrame calls so far (new or overwriting observations alike)

    # `location_3d` and `frames` are plain attributes in the reference (point.py:8-9); here they are properties over
    # slots so that rebinding either one is seen by the maps holding the point (their mirrors re-verify on next use).
    @property
    def location_3d(self):
        return self._loc

    @location_3d.setter
    def location_3d(self, value):
        self._loc = value
        for c in self._cells:
            c[0] += 1

    @property
    def frames(self):
        for c in self._cells:
            if c[2] is not None:  # a holding map has observation batches it has not written into the Point objects yet
                c[2]()
        return self._frames

    @frames.setter
    def frames(self, value):
        self._frames = value
        self._rev += 1
        for c in self._cells:
            c[1] += 1

    def _adopt(self, cell):
        """Called by a Map that starts holding this point."""
        cells = self._cells
        if not cells:
            self._cells = (cell,)
        elif all(c is not cell for c in cells):
            self._cells = cells + (cell,)
            for c in self._cells:   # (sticky) these maps share a point: a bulk write-back in one must tell the other
                if len(c) > 3:
                    c[3] = True

    def __copy__(self):
        q = Point.__new__(Point)
        q.ID, q._frames, q._loc, q._rev, q._cells = self.ID, self.frames, self._loc, self._rev, ()
        return q

    def __deepcopy__(self, memo):
        import copy
        q = Point.__new__(Point)
        memo[id(self)] = q
        q.ID, q._rev, q._cells = copy.deepcopy(self.ID, memo), self._rev, ()
        q._loc = copy.deepcopy(self._loc, memo)
        q._frames = copy.deepcopy(self.frames, memo)
        return q

    # ---- identity / geometry -------------------------------------------------------------------- point.py:11-12,52-56
    def GetID(self):
        """Id under which the map stores this point."""
        return self.ID

    def Get3dPoint(self):
        """Current position estimate (whatever object was stored last: list, ndarray ...)."""
        return self.location_3d

    def UpdatePoint(self, new_location):
        """Rebinds the position (BA write-back); the maps holding the point are told through their change counters."""
        self.location_3d = new_location

    def GetVectorNorm(self):
        """Distance from the world origin (used for the median-depth normalisation, LocalBA.py:178-183)."""
        return np.linalg.norm(self.location_3d)

    # ---- observations --------------------------------------------------------------------------------- point.py:14-50
    def AddFrame(self, frame, uv, descriptor):
        """Records (or overwrites) the observation of this point in `frame`, keyed by the frame's current id."""
        cells = self._cells
        for c in cells:   # (what the `frames` property does: a holding map first writes its pending observation batches)
            if c[2] is not None:
                c[2]()
        self._frames[frame.GetID()] = (frame, uv, descriptor)
        self._rev += 1
        for c in cells:
            c[1] += 1

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
