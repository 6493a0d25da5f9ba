"""Frame / FeatureExtractor / FeatureMatcher with the signatures of the reference's src/v2/frame.py, backed by the HIP
kernels of libvslam_hip.so (no cv2).

Deviations that BASELINE.json's north_star prescribes: the detector is FAST-9/16 + 3x3 NMS (border 15,
threshold 20, strongest 3000) instead of goodFeaturesToTrack, the descriptor is BRIEF-256 instead of SIFT, and the
matcher uses the Hamming norm.  Everything else -- argument order, return tuples, attribute names -- follows the
reference line by line (cited per method).
"""
from collections.abc import Sequence
from functools import partial as _partial
from operator import itemgetter as _itemgetter

import numpy as np

from .context import default_context


def imread(path):
    """cv2.imread replacement (reference frame.py:54-55): 8-bit, 3 channels, B-G-R order; None if unreadable."""
    from PIL import Image
    try:
        im = Image.open(path)
        if im.mode in ("I;16", "I;16B", "I"):  # cv2.imread without flags converts 16-bit depth to 8-bit BGR
            a = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)
            return np.ascontiguousarray(np.repeat(a[:, :, None], 3, axis=2))
        a = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(a[:, :, ::-1])
    except (OSError, ValueError):
        return None


class _PyDMatch(tuple):
    """The fields of cv2.DMatch the reference reads (frame.py:33-47, main.py:187-188,210): queryIdx, trainIdx, imgIdx,
    distance.  Pure-Python stand-in for visual_slam_amd._rows.DMatch (the C type build() compiles): a tuple underneath
    with C-level accessors."""
    __slots__ = ()
    queryIdx = property(_itemgetter(0))
    trainIdx = property(_itemgetter(1))
    imgIdx = property(_itemgetter(2))
    distance = property(_itemgetter(3))

    def __new__(cls, queryIdx, trainIdx, distance, imgIdx=0):
        return tuple.__new__(cls, (int(queryIdx), int(trainIdx), int(imgIdx), float(distance)))

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%g)" % (self[0], self[1], self[3])


def _py_rows(query_idx, train_idx, distance):
    n = len(query_idx)
    rows = zip(query_idx.tolist(), train_idx.tolist(), [0] * n, distance.astype(np.float64).tolist())
    return [[m] for m in map(_partial(tuple.__new__, _PyDMatch), rows)]


try:  # the match rows built in C (visual_slam_amd/cext/_rows.c): 130 -> ~40 us per frame of the class-API tracking loop
    from ._rows import DMatch, rows as _match_rows
except ImportError:  # not built (a source checkout before __graft_entry__.build()): same behaviour, slower
    DMatch, _match_rows = _PyDMatch, _py_rows


class MatchList(Sequence):
    """`matches` as the reference returns it -- a list of one-element lists [[DMatch], ...] (frame.py:47) -- built
    lazily from the arrays the GPU produced.  `m[0].queryIdx` / iteration / len() behave like the reference's list;
    array consumers can use .query_idx / .train_idx / .distance directly and skip the Python objects."""

    def __init__(self, query_idx, train_idx, distance):
        self.query_idx = np.ascontiguousarray(query_idx, np.int32)
        self.train_idx = np.ascontiguousarray(train_idx, np.int32)
        self.distance = np.ascontiguousarray(distance, np.int32)
        self._rows = None

    def __len__(self):
        return int(self.query_idx.shape[0])

    def __iter__(self):
        # the reference's callers iterate `for m in matches: m[0].queryIdx` (main.py:187-188,210): all rows are built in one
        # call, once (the reference's `matches` is a real list: iterating it twice yields the same objects)
        if self._rows is None:
            self._rows = _match_rows(self.query_idx, self.train_idx, self.distance)
        return iter(self._rows)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self._rows is not None:
            return self._rows[i]
        return [DMatch(int(self.query_idx[i]), int(self.train_idx[i]), float(self.distance[i]))]


class FeatureExtractor:
    """reference frame.py:5-14"""

    def __init__(self, threshold=20, max_keypoints=3000, context=None):
        self.threshold = int(threshold)          # cv2.ORB's FAST threshold
        self.max_keypoints = int(max_keypoints)  # the reference asks goodFeaturesToTrack for 3000 (frame.py:11)
        self._ctx = context
        self.last_scores = None

    def compute_features(self, img):
        """img: BGR uint8 [H, W, 3] -> (keypoints float32 [N, 2] as cv2.KeyPoint_convert gives them, descriptors uint8
        [N, 32]).  One upload, two kernel launches (vs_detect_describe_bgr)."""
        ctx = self._ctx or default_context()
        own = ctx._track_owner
        if own is not None:
            # a local map's tracking period is resident on this context (map.py, _PeriodMirror): the frame is detected,
            # described AND matched against the key frame there, in one chain with one synchronisation; match_features
            # below recognises the pair and hands the matches out without touching the GPU again
            r = own.speculate_front(img, self.threshold, self.max_keypoints)
            if r is not None:
                self.last_scores = None
                return r
        xy, score, desc = ctx.detect_describe_bgr(img, self.threshold, self.max_keypoints)
        self.last_scores = score
        return xy, desc


class FeatureMatcher:
    """reference frame.py:16-49"""

    def __init__(self, context=None):
        self._ctx = context

    def match_features(self, kp1, desc1, kp2, desc2, ratio=0.8):
        """knnMatch(desc1, desc2, k=2) + Lowe's ratio test, survivors in query order.
        Returns (matches, pts1, ft1, pts2, ft2) exactly as frame.py:49."""
        ctx = self._ctx or default_context()
        own = ctx._track_owner
        if own is not None and own.spec is not None:
            hit = own.spec_matches(kp1, desc1, kp2, desc2, ratio)
            if hit is not None:  # key frame x the frame in flight: the resident period matched them already
                mq, mt, md = hit
                return MatchList(mq, mt, md), kp1[mq], desc1[mq], kp2[mt], desc2[mt]
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        if desc1.shape[0] > 0 and desc2.shape[0] < 2:
            # the reference's `for m, n in rawMatches` cannot unpack a single neighbour (frame.py:30)
            raise ValueError("not enough values to unpack (expected 2, got %d)" % desc2.shape[0])
        if desc1.shape[0] == 0:
            mq = mt = md = np.zeros(0, np.int32)
        else:
            mq, mt, md = ctx.match_ratio(desc1, desc2, ratio)
        kp1 = np.asarray(kp1)
        kp2 = np.asarray(kp2)
        return MatchList(mq, mt, md), kp1[mq], desc1[mq], kp2[mt], desc2[mt]


class Frame:
    """reference frame.py:51-125.  rgb_fp / d_path may also be arrays (BGR uint8 image / anything) so that a caller
    with frames already in memory does not go through files."""

    def __init__(self, rgb_fp, d_path, id):
        self.rgb = rgb_fp if isinstance(rgb_fp, np.ndarray) else imread(rgb_fp)
        self._d_path = d_path
        self._depth = d_path if isinstance(d_path, np.ndarray) else None
        self.keypoints, self.features = None, None
        self.ID = id
        self.pose = None
        self.parents = {}
        self.childs = []
        self.keyframe = False

    @property
    def depth(self):
        # the reference decodes the depth image eagerly and never reads it; decode on first use
        if self._depth is None and self._d_path is not None:
            self._depth = imread(self._d_path)
        return self._depth

    @depth.setter
    def depth(self, value):
        self._depth = value

    def ClearParent(self):
        self.parents = {}

    def AddParent(self, parent_frame_id, transition):
        self.parents[parent_frame_id] = transition

    def GetParentIDs(self):
        return self.parents.keys()

    def GetTransitionWithParentID(self, parent_id):
        return self.parents[parent_id]

    def process_frame(self, feature_extractor):
        self.keypoints, self.features = self.feature_extract(self.rgb, feature_extractor)
        return self.keypoints, self.features, self.rgb

    def feature_extract(self, rgb, feature_extractor):
        return feature_extractor.compute_features(rgb)

    def AddPose(self, init_pose):
        self.pose = init_pose

    def UpdatePose(self, new_pose):
        self.pose = new_pose

    def AddChild(self, child_frame):
        self.childs.append(child_frame)

    def GetPose(self):
        return self.pose

    def SetAsKeyFrame(self):
        self.keyframe = True

    def GetKeyPoints(self):
        return self.keypoints

    def GetFeatures(self):
        return self.features

    def GetID(self):
        return self.ID

    def IsKeyFrame(self):
        return self.keyframe

    def AddID(self, new_id):
        self.ID = new_id
