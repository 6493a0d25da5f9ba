"""ctypes binding of libvslam_hip.so (include/vslam_hip.h).  There is NO CPU fallback: if the library is missing,
or no MI355X is present, the calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VS_LIB_PATH") or os.path.join(_HERE, "libvslam_hip.so")  # VS_LIB_PATH: dev builds (A/B variants)
_LIB = None

VS_OK, VS_EINVAL, VS_ENOMEM, VS_EHIP, VS_ENOTPD, VS_ECAP, VS_ENCCL = 0, -1, -2, -3, -4, -5, -6

# Typed pointer tags.  A pointer parameter is declared as a subclass of c_void_p that names its pointee type: ctypes then
# still takes a plain address (ndarray.ctypes.data -- a typed pointer object built with data_as / cast costs ~3 us apiece
# and the tracking loop makes some twenty per frame), while (a) ptr() below refuses an array whose dtype is not the
# declared pointee, and (b) tests/test_abi.py compares every tag with the pointee type in include/vslam_hip.h, so a
# signature drift is caught without a GPU.
def _tag(name, dtype, ctype):
    return type(name, (C.c_void_p,), {"dtype": np.dtype(dtype) if dtype is not None else None, "ctype": ctype})


c_u8p = _tag("c_u8p", np.uint8, "uint8_t")
c_i32p = _tag("c_i32p", np.int32, "int32_t")
c_f32p = _tag("c_f32p", np.float32, "float")
c_f64p = _tag("c_f64p", np.float64, "double")
c_intp = _tag("c_intp", np.intc, "int")           # int* out-parameters (passed with ctypes.byref)
c_ctxp = _tag("c_ctxp", None, "vs_ctx")           # vs_ctx*
c_voidp = _tag("c_voidp", None, "void")           # device pointers, streams, events, communicators


class VsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvslam_hip: %s (status %d)" % (msg, code))
        self.code = code


class BAProblem(C.Structure):
    _fields_ = [
        ("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("n_scale", C.c_int32),
        ("poses", c_f64p), ("pose_fixed", c_u8p), ("points", c_f64p), ("point_fixed", c_u8p),
        ("obs_pose", c_i32p), ("obs_point", c_i32p), ("obs_uv", c_f64p), ("obs_info", c_f64p),
        ("scale_parent", c_i32p), ("scale_child", c_i32p), ("scale_meas", c_f64p),
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("huber_delta", C.c_double), ("dcs_phi", C.c_double),
        ("max_iterations", C.c_int32), ("reserved", C.c_int32),
    ]


class BAResult(C.Structure):
    _fields_ = [
        ("poses_out", c_f64p), ("points_out", c_f64p), ("chi2_trace", c_f64p), ("lambda_trace", c_f64p),
        ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
        ("iterations", C.c_int32), ("trials", C.c_int32), ("not_pd", C.c_int32), ("terminated", C.c_int32),
        ("trial_trace", c_f64p), ("trial_trace_cap", C.c_int32), ("reserved", C.c_int32),
    ]


# name -> (restype, argtypes); every entry point declared in include/vslam_hip.h
SIGNATURES = {
    "vs_abi_version": (C.c_int, []),
    "vs_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "vs_destroy": (C.c_int, [c_ctxp]),
    "vs_last_error": (C.c_char_p, [c_ctxp]),
    "vs_stream": (C.c_void_p, [c_ctxp]),
    "vs_aux_stream": (C.c_void_p, [c_ctxp, C.c_int]),
    "vs_synchronize": (C.c_int, [c_ctxp]),
    "vs_host_alloc": (C.c_int, [c_ctxp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vs_host_free": (C.c_int, [c_ctxp, c_voidp]),
    "vs_gray_mean3_u8": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, c_u8p]),
    "vs_fast9_detect": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                                  c_u8p, c_intp]),
    "vs_brief256": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int, c_u8p, c_i32p, c_intp]),
    "vs_detect_describe_bgr": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p,
                                         c_u8p, c_u8p, c_intp]),
    "vs_detect_describe_bgr_dev": (C.c_int, [c_ctxp, c_voidp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             c_voidp, c_voidp, c_voidp, c_voidp, c_voidp]),
    "vs_hamming_knn2": (C.c_int, [c_ctxp, c_u8p, C.c_int, c_u8p, C.c_int, c_i32p, c_i32p]),
    "vs_match_ratio": (C.c_int, [c_ctxp, c_u8p, C.c_int, c_u8p, C.c_int, C.c_double, c_i32p, c_i32p, c_i32p,
                                 c_intp]),
    "vs_hamming_knn2_dev": (C.c_int, [c_ctxp, c_voidp, C.c_int, c_voidp, C.c_int, c_voidp, c_voidp, c_voidp]),
    "vs_hamming_knn2_packed_dev": (C.c_int, [c_ctxp, c_voidp, C.c_int, c_voidp, C.c_int, c_voidp, c_voidp]),
    "vs_match_ratio_dev": (C.c_int, [c_ctxp, c_voidp, C.c_int, c_voidp, C.c_int, C.c_double, c_voidp, c_voidp, c_voidp,
                                     c_voidp, c_voidp]),
    "vs_triangulate_dlt": (C.c_int, [c_ctxp, c_f64p, c_f64p, c_f64p, c_f64p, C.c_int, C.c_int, c_f64p, c_f64p, c_f64p,
                                     c_f64p]),
    "vs_pnp_ransac": (C.c_int, [c_ctxp, c_f64p, c_f64p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, c_f64p,
                                C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_int, c_f64p, c_i32p, c_intp, c_intp]),
    "vs_essential_ransac": (C.c_int, [c_ctxp, c_f64p, c_f64p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint64,
                                      c_f64p, c_u8p, c_intp, c_intp]),
    "vs_recover_pose": (C.c_int, [c_ctxp, c_f64p, c_f64p, c_f64p, C.c_int, C.c_double, c_f64p, c_f64p, c_u8p, c_f64p,
                                  c_intp]),
    "vs_track_begin": (C.c_int, [c_ctxp, c_f64p, c_u8p, C.c_int, c_f64p, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, C.c_int]),
    "vs_track_frame": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                 C.c_uint64, C.c_int, C.c_double, c_f64p, c_intp, c_intp, c_intp, c_f32p, c_u8p, c_intp,
                                 c_i32p, c_i32p]),
    "vs_track_last_frame": (C.c_int, [c_ctxp, c_f32p, c_u8p, c_intp, c_i32p, c_i32p, c_intp]),
    "vs_track_frame_pipelined": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_uint64, C.c_int, C.c_double, c_intp, c_f64p, c_intp, c_intp,
                                           c_intp, c_f32p, c_u8p, c_intp, c_i32p, c_i32p]),
    "vs_track_push_frame": (C.c_int, [c_ctxp, c_i32p, c_f64p, C.c_int, c_f64p, C.c_int, C.c_double, c_f64p, c_intp]),
    "vs_track_front": (C.c_int, [c_ctxp, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_f32p, c_u8p, c_intp, c_i32p,
                                 c_i32p, c_i32p, c_intp]),
    "vs_track_back_begin": (C.c_int, [c_ctxp, C.c_double, C.c_double, C.c_uint64, C.c_int, C.c_double, c_f64p, C.c_int, c_intp,
                                      c_f64p, c_i32p, c_intp]),
    "vs_track_back_end": (C.c_int, [c_ctxp, c_f64p, c_intp]),
    "vs_track_end": (C.c_int, [c_ctxp]),
    "vs_ba_solve": (C.c_int, [c_ctxp, C.POINTER(BAProblem), C.POINTER(BAResult)]),
    "vs_ba_debug_cholesky": (C.c_int, [c_ctxp, c_f64p, C.c_int, c_f64p, c_f64p, c_intp]),
    "vs_match_status": (C.c_int, [c_ctxp]),
    "vs_hamming_knn2_sharded_dev": (C.c_int, [c_ctxp, c_voidp, C.c_int, c_voidp, C.c_int, c_voidp, C.c_int,
                                              C.c_int, C.c_int, c_voidp, c_voidp, c_voidp, c_voidp, c_voidp]),
}

# test / sweep / profiling hooks (per context, not part of the stable ABI): declared in include/vslam_hip_dev.h
HOOKS = {
    "vs_tune_match": (C.c_int, [c_ctxp, C.c_int, C.c_int]),                 # target workgroups, train staging (-1: keep)
    "vs_tune_ba_structure": (C.c_int, [c_ctxp, C.c_int]),
    "vs_ba_structure_on_device": (C.c_int, [c_ctxp]),
    "vs_ba_last_path": (C.c_int, [c_ctxp, c_intp]),
    "vs_tune_ba_graph": (C.c_int, [c_ctxp, C.c_int, c_f64p]),
    "vs_tune_ba_solve": (C.c_int, [c_ctxp, C.c_int]),
    "vs_tune_ba": (C.c_int, [c_ctxp, C.c_int, C.c_int, C.c_int, C.c_int]),  # schur variant, points / workgroup, slab cap, motion variant
    "vs_match_profile": (C.c_int, [c_ctxp, C.c_int]),
    "vs_match_profile_read": (C.c_int, [c_ctxp, C.POINTER(C.c_float)]),
    "vs_match_stamps": (C.c_int, [c_ctxp, C.c_int]),
    "vs_match_stamps_read": (C.c_int, [c_ctxp, c_f64p, C.c_int]),
    "vs_mo_profile": (C.c_int, [c_ctxp, C.c_int]),
    "vs_mo_profile_read": (C.c_int, [c_ctxp, c_f64p, C.c_int]),
    "vs_match_debug_raise": (C.c_int, [c_ctxp]),
    "vs_track_debug": (C.c_int, [c_ctxp, C.c_int, c_intp]),
    "vs_debug_poison_alloc": (C.c_int, [c_ctxp, C.c_int]),
    "vs_pnp_profile": (C.c_int, [c_ctxp, C.c_int]),
    "vs_pnp_profile_read": (C.c_int, [c_ctxp, c_f64p, C.c_int]),
}


def load():
    """Load libvslam_hip.so.  Raises ImportError with build instructions when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C visual_slam_amd/csrc` (or __graft_entry__.build()). "
                          "There is no CPU fallback for the tracking hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:  # reported by tests/test_abi.py; calling it raises AttributeError
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    lib._vs_missing = missing
    for name, (res, args) in HOOKS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _LIB = lib
    return lib


def device_count():
    """Number of HIP devices, without creating a context (0 in the CPU-only build container)."""
    try:
        hip = C.CDLL("libamdhip64.so")
    except OSError:
        return 0
    n = C.c_int(0)
    rc = hip.hipGetDeviceCount(C.byref(n))
    return n.value if rc == 0 else 0


def ptr(a, t):
    """Address of a C-contiguous ndarray for a parameter whose pointee type is the tag `t`; the caller keeps `a` alive.
    A wrong element type or a strided array would be silent memory corruption behind a void*: refused here."""
    if a.dtype is not t.dtype and a.dtype != t.dtype:
        raise TypeError("libvslam_hip: %s* parameter got an array of %s" % (t.ctype, a.dtype))
    if not a.flags.c_contiguous:
        raise TypeError("libvslam_hip: %s* parameter needs a C-contiguous array" % t.ctype)
    return a.ctypes.data


def as_u8(a, shape_tail=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape_tail is not None:
        a = a.reshape((-1,) + tuple(shape_tail))
    return a
