"""Frame / FeatureExtractor / FeatureMatcher with the signatures of the reference's src/v2/frame.py, backed by the HIP
kernels of libvslam_hip.so (no cv2).

Deviations that BASELINE.json's north_star prescribes: the detector is FAST-9/16 + 3x3 NMS (border 15,
threshold 20, strongest 3000) instead of goodFeaturesToTrack, the descriptor is BRIEF-256 instead of SIFT, and the
matcher uses the Hamming norm.  Everything else -- argument order, return tuples, attribute names -- follows the
reference line by line (cited per method).
"""
from collections.abc import Sequence

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


class DMatch:
    """The fields of cv2.DMatch the reference reads (frame.py:33-47, main.py:187-188,210)."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%g)" % (self.queryIdx, self.trainIdx, self.distance)


class MatchList(Sequence):
    """`matches` as the reference returns it -- a list of one-element lists [[DMatch], ...] (frame.py:47) -- built
    lazily from the arrays the GPU produced.  `m[0].queryIdx` / iteration / len() behave like the reference's list;
    array consumers can use .query_idx / .train_idx / .distance directly and skip the Python objects."""

    def __init__(self, query_idx, train_idx, distance):
        self.query_idx = np.asarray(query_idx, np.int32)
        self.train_idx = np.asarray(train_idx, np.int32)
        self.distance = np.asarray(distance, np.int32)

    def __len__(self):
        return int(self.query_idx.shape[0])

    def __iter__(self):
        # the reference's callers iterate `for m in matches: m[0].queryIdx` (main.py:187-188,210): build the objects
        # from plain Python scalars in one pass instead of one NumPy scalar conversion per attribute
        new = DMatch.__new__
        for q, t, d in zip(self.query_idx.tolist(), self.train_idx.tolist(), self.distance.tolist()):
            m = new(DMatch)
            m.queryIdx = q
            m.trainIdx = t
            m.imgIdx = 0
            m.distance = float(d)
            yield [m]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return [DMatch(self.query_idx[i], self.train_idx[i], self.distance[i])]


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
