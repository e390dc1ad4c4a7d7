"""ctypes binding of libdfx.so (the C ABI declared in include/dfx.h).

There is NO CPU fallback: if the shared library is missing or no gfx950 device is usable the calls raise.
``import torch`` must happen before the library is loaded so that libdfx.so binds to the same
libamdhip64.so.7 instance PyTorch uses (device pointers and streams are then shared).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFX_LIB") or os.path.join(_HERE, "libdfx.so")   # DFX_LIB: tuning variants (tools/variants.sh)

DFX_OK = 0
DFX_SCHEDULE_AUTO, DFX_SCHEDULE_STATIC, DFX_SCHEDULE_DYNAMIC = 0, 1, 2
DFX_E_INVALID = -1
DFX_E_HIP = -2
DFX_E_NOGPU = -3
DFX_MFMA_F32_CHAIN = 0
DFX_MFMA_BF16X3 = 1
DFX_MFMA_AUTO = 2
DFX_COMM_ID_BYTES = 128
DFX_WAIT_STREAM, DFX_WAIT_POLL = 0, 1
DFX_OPT_SIMPLE_DESC_ZEROCOPY, DFX_OPT_STEP_DESC_ZEROCOPY = 1, 2


class DfxError(RuntimeError):
    """Raised for every non-zero status of the C ABI (mirrors vc::CUDAException thrown by
    CudaCheckLastError in the reference, sources/cuda/launch_utils.h:26-32)."""

    def __init__(self, code, msg):
        super().__init__(f"dfx error {code}: {msg}")
        self.code = code


class Img(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch_bytes", C.c_size_t), ("w", C.c_uint32), ("h", C.c_uint32)]


class SE3(C.Structure):
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3)]


class Cam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("u0", C.c_float), ("v0", C.c_float), ("w", C.c_float), ("h", C.c_float)]


class SfmParams(C.Structure):
    _fields_ = [("huber_delta", C.c_float), ("avg_dpt", C.c_float), ("min_dpt", C.c_float), ("valid_border", C.c_int32), ("step_blocks", C.c_int32)]


class CorrItem(C.Structure):
    _fields_ = [("residual", C.c_float), ("_pad", C.c_uint32), ("inliers", C.c_uint64)]


class TrackLevel(C.Structure):
    _fields_ = [("cam", Cam), ("img0", Img), ("img1", Img), ("dpt0", Img), ("grad1", Img), ("iterations", C.c_int32)]


class TrackResult(C.Structure):
    _fields_ = [("pose_ck", SE3), ("inliers_frac", C.c_float), ("error", C.c_float), ("residual", C.c_float), ("inliers", C.c_uint64),
                ("iterations", C.c_int32), ("solver_failures", C.c_int32)]


class SfmPair(C.Structure):
    _fields_ = [("pose0", SE3), ("pose1", SE3), ("cam", Cam), ("img0", Img), ("img1", Img), ("dpt0", Img),
                ("valid0", Img), ("prx0_jac", Img), ("grad1", Img)]


class SE3Pair(C.Structure):
    _fields_ = [("pose_10", SE3), ("cam", Cam), ("img0", Img), ("img1", Img), ("dpt0", Img), ("grad1", Img)]


DFX_MAX_PYR_LEVELS = 8


class Pyramid(C.Structure):
    _fields_ = [("levels", C.c_int32), ("img", Img * DFX_MAX_PYR_LEVELS), ("grad", Img * DFX_MAX_PYR_LEVELS)]


class SparseGeoFactor(C.Structure):
    _fields_ = [("pose0", SE3), ("pose1", SE3), ("cam", Cam), ("code0", C.POINTER(C.c_float)), ("code1", C.POINTER(C.c_float)), ("points_xy", C.c_void_p),
                ("n_points", C.c_int32), ("points_on_device", C.c_int32), ("prx0_orig", Img), ("prx0_jac", Img), ("prx1_orig", Img), ("prx1_jac", Img),
                ("dpt1_grad", Img)]


def item_jtj_len(np_):
    return np_ * (np_ + 1) // 2


def item_inliers_offset(np_):
    return ((item_jtj_len(np_) + np_ + 1) * 4 + 7) & ~7


def item_size(np_):
    return item_inliers_offset(np_) + 8


_lib = None

_PROTOS = {
    "dfx_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "dfx_ctx_destroy": (None, [C.c_void_p]),
    "dfx_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfx_ctx_device": (C.c_int, [C.c_void_p]),
    "dfx_set_tail_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfx_tail_join": (C.c_int, [C.c_void_p]),
    "dfx_last_error": (C.c_char_p, []),
    "dfx_version": (C.c_char_p, []),
    "dfx_sync": (C.c_int, [C.c_void_p]),
    "dfx_sfm_set_step_blocks": (C.c_int, [C.c_void_p, C.c_int]),
    "dfx_sfm_auto_step_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "dfx_device_cu_count": (C.c_int, [C.c_void_p]),
    "dfx_set_mfma_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "dfx_last_mfma_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dfx_debug_read_valid0_shadow": (C.c_int, [C.c_void_p, C.POINTER(Img), C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]),
    "dfx_set_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "dfx_set_result_wait": (C.c_int, [C.c_void_p, C.c_int]),
    "dfx_ctx_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "dfx_last_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dfx_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "dfx_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "dfx_profile_read_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dfx_debug_read_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfx_debug_pyramid_launch_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfx_img_alloc": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.POINTER(Img)]),
    "dfx_img_free": (C.c_int, [C.c_void_p, C.POINTER(Img)]),
    "dfx_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dfx_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfx_img_upload": (C.c_int, [C.c_void_p, C.POINTER(Img), C.c_void_p, C.c_size_t, C.c_size_t]),
    "dfx_img_download": (C.c_int, [C.c_void_p, C.POINTER(Img), C.c_void_p, C.c_size_t, C.c_size_t]),
    "dfx_img_fill_f32": (C.c_int, [C.c_void_p, C.POINTER(Img), C.c_float]),
    "dfx_se3_step": (C.c_int, [C.c_void_p, C.POINTER(SE3), C.POINTER(Cam), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img),
                               C.POINTER(Img), C.c_float, C.c_void_p]),
    "dfx_se3_warp": (C.c_int, [C.c_void_p, C.POINTER(SE3), C.POINTER(Cam), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img),
                               C.POINTER(Img), C.POINTER(CorrItem)]),
    "dfx_se3_step_batch_async": (C.c_int, [C.c_void_p, C.POINTER(SE3Pair), C.c_int, C.c_float, C.c_void_p]),
    "dfx_se3_step_batch": (C.c_int, [C.c_void_p, C.POINTER(SE3Pair), C.c_int, C.c_float, C.c_void_p]),
    "dfx_sfm_error_batch_async": (C.c_int, [C.c_void_p, C.POINTER(SfmParams), C.POINTER(SfmPair), C.c_int, C.c_void_p]),
    "dfx_sfm_error_batch": (C.c_int, [C.c_void_p, C.POINTER(SfmParams), C.POINTER(SfmPair), C.c_int, C.POINTER(CorrItem)]),
    "dfx_track_frame": (C.c_int, [C.c_void_p, C.POINTER(SE3), C.POINTER(TrackLevel), C.c_int, C.c_float, C.POINTER(TrackResult)]),
    "dfx_track_frame_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SE3), C.POINTER(TrackLevel), C.c_int, C.c_float, C.POINTER(TrackResult)]),
    "dfx_sparse_geometric_linearize": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SE3), C.POINTER(SE3), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                 C.POINTER(Cam), C.POINTER(C.c_int32), C.c_int, C.POINTER(Img), C.POINTER(Img), C.POINTER(Img),
                                                 C.POINTER(Img), C.POINTER(Img), C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "dfx_sparse_geometric_linearize_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SparseGeoFactor), C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dfx_sparse_geometric_linearize_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SparseGeoFactor), C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "dfx_sparse_geometric_gram_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SparseGeoFactor), C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "dfx_sparse_geometric_gram_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SparseGeoFactor), C.c_int, C.c_float, C.c_float, C.POINTER(C.c_float)]),
    "dfx_sfm_step": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SE3), C.POINTER(SE3), C.POINTER(Cam), C.POINTER(SfmParams),
                               C.POINTER(Img), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img),
                               C.POINTER(Img), C.c_void_p]),
    "dfx_sfm_error": (C.c_int, [C.c_void_p, C.POINTER(SE3), C.POINTER(SE3), C.POINTER(Cam), C.POINTER(SfmParams),
                                C.POINTER(Img), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img), C.POINTER(Img),
                                C.POINTER(CorrItem)]),
    "dfx_sfm_step_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SfmParams), C.POINTER(SfmPair), C.c_int, C.c_void_p]),
    "dfx_sfm_step_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SfmParams), C.POINTER(SfmPair), C.c_int, C.c_void_p]),
    "dfx_graph_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "dfx_graph_destroy": (None, [C.c_void_p]),
    "dfx_graph_system_floats": (C.c_size_t, [C.c_void_p]),
    "dfx_graph_assemble_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dfx_sfm_step_batch_assemble_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SfmParams), C.POINTER(SfmPair), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dfx_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "dfx_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dfx_comm_destroy": (None, [C.c_void_p]),
    "dfx_comm_rank": (C.c_int, [C.c_void_p]),
    "dfx_comm_world": (C.c_int, [C.c_void_p]),
    "dfx_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfx_graph_reduce_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dfx_comm_reduce_f32_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "dfx_items_all_gather_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dfx_comm_broadcast_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "dfx_update_depth": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(Img), C.POINTER(Img), C.c_float,
                                   C.POINTER(Img)]),
    "dfx_update_depth_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(Img), C.POINTER(Img), C.c_float, C.POINTER(Img)]),
    "dfx_sfm_linearize_batch_async": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SfmParams), C.POINTER(SfmPair), C.POINTER(Img), C.POINTER(C.c_float), C.c_int,
                                                C.c_void_p]),
    "dfx_sfm_linearize_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SfmParams), C.POINTER(SfmPair), C.POINTER(Img), C.POINTER(C.c_float), C.c_int,
                                          C.c_void_p]),
    "dfx_sobel_gradients": (C.c_int, [C.c_void_p, C.POINTER(Img), C.POINTER(Img)]),
    "dfx_gaussian_blur_down": (C.c_int, [C.c_void_p, C.POINTER(Img), C.POINTER(Img)]),
    "dfx_build_pyramid_batch_async": (C.c_int, [C.c_void_p, C.POINTER(Pyramid), C.c_int]),
    "dfx_build_pyramid": (C.c_int, [C.c_void_p, C.POINTER(Pyramid)]),
    "dfx_squared_error": (C.c_int, [C.c_void_p, C.POINTER(Img), C.POINTER(Img), C.POINTER(C.c_float)]),
    "dfx_depth_aligner_step": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(Img), C.POINTER(Img),
                                         C.POINTER(Img), C.c_float, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())


def lib():
    """Loads libdfx.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DfxError(DFX_E_NOGPU, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                        "or `make -C deepfactors_amd/csrc`")
        try:
            import torch  # noqa: F401  (binds libamdhip64.so.7 first, see module docstring)
        except Exception:  # pragma: no cover - torch is only plumbing
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != DFX_OK:
        raise DfxError(rc, lib().dfx_last_error().decode("utf-8", "replace"))
