"""visual_slam_amd -- MI355X-native per-frame tracking hot path behind the src/v2 API of juuso-oskari/visual_slam.

Python host code (Frame / Point / Map / FeatureExtractor / FeatureMatcher / BundleAdjustment, same names and
signatures as the reference's src/v2) over hand-written HIP kernels for gfx950, reached through the C ABI of
libvslam_hip.so (include/vslam_hip.h) with ctypes.  No CPU fallback exists on the product path.
"""
from ._capi import VsError, load as load_library  # noqa: F401
from .context import Context, default_context  # noqa: F401

__all__ = ["Context", "default_context", "VsError", "load_library"]
