"""ctypes loader for liblmx.so (C ABI: include/lmx.h).  No fallback: a missing library is an error."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("LMX_SO_PATH") or os.path.join(CSRC, "liblmx.so")  # override: A/B runs of kernel variants

# every symbol include/lmx.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "lmx_default_normal_lut", "lmx_bank_set_normal_lut", "lmx_bank_get_normal_lut", "lmx_bank_load_normal_lut", "lmx_bank_normal_lut_origin", "lmx_bank_require_normal_lut",
    "lmx_bank_clone", "lmx_bank_fingerprint", "lmx_bank_load_yaml_cached", "lmx_bank_release", "lmx_bank_save_binary", "lmx_bank_load_binary", "lmx_ctx_acquire", "lmx_ctx_unref", "lmx_cache_trim", "lmx_ctx_lock", "lmx_ctx_unlock",
    "lmx_yaml_open", "lmx_yaml_close", "lmx_yaml_root", "lmx_yaml_kind", "lmx_yaml_scalar", "lmx_yaml_size", "lmx_yaml_item", "lmx_yaml_key", "lmx_yaml_get",
    "lmx_group_unique_id", "lmx_group_create", "lmx_group_destroy", "lmx_group_size", "lmx_group_frame_groups", "lmx_merge_gathered_groups", "lmx_group_gather_capacity", "lmx_group_match_batch",
    "lmx_group_upload", "lmx_group_submit", "lmx_group_finish", "lmx_group_depth", "lmx_group_collective_name", "lmx_ctx_export_oldest_on",
    "lmx_bank_create", "lmx_bank_add_class", "lmx_bank_add_template", "lmx_bank_load_yaml", "lmx_bank_save_yaml", "lmx_bank_destroy",
    "lmx_bank_pyramid_levels", "lmx_bank_T", "lmx_bank_num_modalities", "lmx_bank_modality", "lmx_bank_num_classes",
    "lmx_bank_class_id", "lmx_bank_num_templates", "lmx_bank_get_template",
    "lmx_ctx_create", "lmx_ctx_destroy", "lmx_match", "lmx_match_batch", "lmx_ctx_upload", "lmx_ctx_upload_masks", "lmx_match_masked", "lmx_ctx_upload_wait", "lmx_host_alloc", "lmx_host_free", "lmx_ctx_upload_raw", "lmx_ctx_enqueue",
    "lmx_ctx_collect", "lmx_ctx_collect_flat", "lmx_ctx_raw_matches", "lmx_merge_raw", "lmx_ctx_export_raw", "lmx_ctx_export_raw_on", "lmx_ctx_release", "lmx_ctx_max_outstanding", "lmx_stream_copy", "lmx_stream_copy_blocks", "lmx_merge_gathered", "lmx_ctx_sync", "lmx_renderer_params_load", "lmx_renderer_params_save", "lmx_renderer_params_free", "lmx_cluster_matches", "lmx_ctx_set_cluster_sidecar", "lmx_ctx_collect_clusters", "lmx_ctx_debug_read", "lmx_debug_orientation_labels", "lmx_debug_introsort_perm", "lmx_debug_introsort_perm_score", "lmx_debug_device_sort_perm", "lmx_ctx_stats",
    "lmx_num_kernels", "lmx_kernel_name", "lmx_ctx_device_kernel_name", "lmx_ctx_set_profiling", "lmx_ctx_kernel_time", "lmx_ctx_reset_profiling",
    "lmx_ctx_algorithmic_bytes", "lmx_last_error", "lmx_version",
]

(LMX_OK, LMX_ERR_INVALID_ARG, LMX_ERR_SHAPE, LMX_ERR_NO_DEVICE, LMX_ERR_HIP, LMX_ERR_OVERFLOW, LMX_ERR_IO,
 LMX_ERR_PARSE, LMX_ERR_NOT_FOUND) = range(9)
LMX_MOD_COLOR_GRADIENT, LMX_MOD_DEPTH_NORMAL = 0, 1
LMX_DBG_QUANTIZED, LMX_DBG_LINEAR_MEMORY, LMX_DBG_PYRAMID_BGR = 0, 1, 2
LMX_NORMAL_LUT_SIZE = 8000
LMX_LUT_DEFAULT, LMX_LUT_USER, LMX_LUT_SIDECAR, LMX_LUT_UNKNOWN = range(4)


class ModalityDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("weak_threshold", C.c_float), ("strong_threshold", C.c_float),
                ("num_features", C.c_int32), ("distance_threshold", C.c_int32), ("difference_threshold", C.c_int32),
                ("extract_threshold", C.c_int32)]


class BankDesc(C.Structure):
    _fields_ = [("pyramid_levels", C.c_int32), ("T", C.POINTER(C.c_int32)), ("n_modalities", C.c_int32),
                ("modalities", C.POINTER(ModalityDesc))]


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("channels", C.c_int32),
                ("elem_size", C.c_int32), ("row_stride_bytes", C.c_size_t)]


class PreDesc(C.Structure):
    _fields_ = [("src_width", C.c_int32), ("src_height", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32),
                ("blur3", C.c_int32), ("mono", C.c_int32), ("depth_float_m", C.c_int32)]


class ClusterParams(C.Structure):
    _fields_ = [("vote_row_col_step", C.c_int32), ("renderer_radius_min", C.c_double), ("renderer_radius_step", C.c_double),
                ("cluster_size_thresh", C.c_int32)]


class RendererParams(C.Structure):
    _fields_ = [("n_templates", C.c_size_t), ("obj_origin_dists", C.POINTER(C.c_double)), ("rects", C.POINTER(C.c_int32)), ("distances", C.POINTER(C.c_double)),
                ("R", C.POINTER(C.c_double)), ("T", C.POINTER(C.c_double)), ("K", C.POINTER(C.c_double)),
                ("renderer_n_points", C.c_int32), ("renderer_angle_step", C.c_int32), ("renderer_width", C.c_int32), ("renderer_height", C.c_int32),
                ("renderer_radius_min", C.c_double), ("renderer_radius_max", C.c_double), ("renderer_radius_step", C.c_double),
                ("renderer_focal_length_x", C.c_double), ("renderer_focal_length_y", C.c_double), ("renderer_near", C.c_double), ("renderer_far", C.c_double)]


class GroupDesc(C.Structure):
    _fields_ = [("n_devices", C.c_int32), ("devices", C.POINTER(C.c_int32)), ("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32),
                ("max_candidates", C.c_int32), ("gather_capacity", C.c_int32), ("flags", C.c_int32), ("unique_id", C.c_void_p), ("rank", C.c_int32),
                ("world", C.c_int32), ("device", C.c_int32), ("collective", C.c_int32), ("frame_groups", C.c_int32)]


class CtxDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("max_batch", C.c_int32),
                ("max_candidates", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
                ("stream", C.c_void_p), ("flags", C.c_int32)]


def build(force=False, verbose=False):
    """Compile liblmx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building liblmx.so failed:\n" + res.stdout)
    if verbose:
        print(res.stdout)
    return SO_PATH


_lib = None


def lib():
    """Load liblmx.so.  Raises (never falls back to a CPU path) when the library is missing or stale symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "liblmx.so not found at %s: the HIP extension is required (there is no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C %s`." % (SO_PATH, CSRC))
    # PyTorch-ROCm bundles its own libamdhip64 under a file name that does not match the SONAME liblmx.so asks for: loaded in the order
    # liblmx -> torch, the process ends up with TWO HIP runtimes and the second one finds no device ("No HIP GPUs are available").  With
    # torch loaded first its runtime satisfies liblmx's dependency and both share one.  So: where torch is installed, load it first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError("liblmx.so lacks symbols declared in include/lmx.h: %s" % missing)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    L.lmx_last_error.restype = C.c_char_p
    L.lmx_version.restype = C.c_char_p
    L.lmx_bank_create.argtypes = [C.POINTER(BankDesc), C.POINTER(vp)]
    L.lmx_bank_add_class.argtypes = [vp, C.c_char_p, C.c_int32, i32p, i32p, C.c_int64]
    L.lmx_bank_add_template.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32, C.c_char_p, C.POINTER(Image), i32p, i32p]
    L.lmx_default_normal_lut.argtypes = [vp]
    L.lmx_bank_set_normal_lut.argtypes = [vp, vp]
    L.lmx_bank_get_normal_lut.argtypes = [vp, vp]
    L.lmx_bank_load_normal_lut.argtypes = [vp, C.c_char_p]
    L.lmx_bank_normal_lut_origin.argtypes = [vp]
    L.lmx_bank_require_normal_lut.argtypes = [vp]
    L.lmx_bank_clone.argtypes = [vp, C.POINTER(vp)]
    L.lmx_bank_fingerprint.argtypes = [vp]
    L.lmx_bank_fingerprint.restype = C.c_uint64
    L.lmx_bank_load_yaml_cached.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lmx_bank_save_binary.argtypes = [vp, C.c_char_p]
    L.lmx_bank_load_binary.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lmx_bank_release.argtypes = [vp]
    L.lmx_bank_release.restype = None
    L.lmx_ctx_acquire.argtypes = [vp, C.POINTER(CtxDesc), C.POINTER(vp), C.POINTER(C.c_int32)]
    L.lmx_ctx_unref.argtypes = [vp]
    L.lmx_cache_trim.argtypes = []
    L.lmx_cache_trim.restype = None
    L.lmx_ctx_unref.restype = None
    for fn in (L.lmx_ctx_lock, L.lmx_ctx_unlock):
        fn.argtypes = [vp]
        fn.restype = None
    L.lmx_yaml_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lmx_yaml_close.argtypes = [vp]
    L.lmx_yaml_close.restype = None
    L.lmx_yaml_root.argtypes = [vp]
    L.lmx_yaml_root.restype = vp
    L.lmx_yaml_kind.argtypes = [vp]
    L.lmx_yaml_scalar.argtypes = [vp]
    L.lmx_yaml_scalar.restype = C.c_char_p
    L.lmx_yaml_size.argtypes = [vp]
    L.lmx_yaml_item.argtypes = [vp, C.c_int32]
    L.lmx_yaml_item.restype = vp
    L.lmx_yaml_key.argtypes = [vp, C.c_int32]
    L.lmx_yaml_key.restype = C.c_char_p
    L.lmx_yaml_get.argtypes = [vp, C.c_char_p]
    L.lmx_yaml_get.restype = vp
    L.lmx_group_unique_id.argtypes = [vp]
    L.lmx_group_create.argtypes = [vp, C.POINTER(GroupDesc), C.POINTER(vp)]
    L.lmx_group_destroy.argtypes = [vp]
    L.lmx_group_destroy.restype = None
    L.lmx_group_size.argtypes = [vp]
    L.lmx_group_gather_capacity.argtypes = [vp]
    L.lmx_group_frame_groups.argtypes = [vp]
    L.lmx_group_match_batch.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_group_upload.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32]
    L.lmx_group_submit.argtypes = [vp, C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32]
    L.lmx_group_finish.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_group_depth.argtypes = [vp]
    L.lmx_group_collective_name.argtypes = [vp]
    L.lmx_group_collective_name.restype = C.c_char_p
    L.lmx_ctx_export_oldest_on.argtypes = [vp, vp, C.c_size_t, vp]
    L.lmx_bank_load_yaml.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.lmx_bank_save_yaml.argtypes = [vp, C.c_char_p]
    L.lmx_bank_destroy.argtypes = [vp]
    L.lmx_bank_destroy.restype = None
    L.lmx_bank_pyramid_levels.argtypes = [vp]
    L.lmx_bank_T.argtypes = [vp, C.c_int32]
    L.lmx_bank_num_modalities.argtypes = [vp]
    L.lmx_bank_modality.argtypes = [vp, C.c_int32, C.POINTER(ModalityDesc)]
    L.lmx_bank_num_classes.argtypes = [vp]
    L.lmx_bank_class_id.argtypes = [vp, C.c_int32]
    L.lmx_bank_class_id.restype = C.c_char_p
    L.lmx_bank_num_templates.argtypes = [vp, C.c_char_p]
    L.lmx_bank_get_template.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, i32p, i32p, i32p, C.POINTER(i32p), i32p]
    L.lmx_ctx_create.argtypes = [vp, C.POINTER(CtxDesc), C.POINTER(vp)]
    L.lmx_ctx_destroy.argtypes = [vp]
    L.lmx_ctx_destroy.restype = None
    L.lmx_match.argtypes = [vp, C.POINTER(Image), C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32, vp, C.c_size_t,
                            C.POINTER(C.c_size_t)]
    L.lmx_match_batch.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32,
                                  vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_ctx_upload.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32]
    L.lmx_ctx_upload_wait.argtypes = [vp]
    L.lmx_ctx_upload_masks.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32]
    L.lmx_match_masked.argtypes = [vp, C.POINTER(Image), C.POINTER(Image), C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.lmx_host_free.argtypes = [vp]
    L.lmx_host_free.restype = None
    L.lmx_ctx_upload_raw.argtypes = [vp, C.c_int32, C.POINTER(Image), C.c_int32, C.POINTER(PreDesc)]
    L.lmx_ctx_enqueue.argtypes = [vp, C.c_int32, C.c_float, C.POINTER(C.c_char_p), C.c_int32]
    L.lmx_ctx_collect.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_ctx_collect_flat.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_ctx_raw_matches.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.lmx_ctx_export_raw.argtypes = [vp, vp, C.c_size_t]
    L.lmx_ctx_export_raw_on.argtypes = [vp, vp, C.c_size_t, vp]
    L.lmx_ctx_release.argtypes = [vp]
    L.lmx_ctx_max_outstanding.argtypes = [vp]
    L.lmx_stream_copy.argtypes = [vp, vp, C.c_size_t, vp]
    L.lmx_stream_copy_blocks.argtypes = [vp, vp, C.c_int32, C.c_size_t, C.c_size_t, vp]
    L.lmx_merge_gathered.argtypes = [vp, C.c_int32, C.c_size_t, C.c_size_t, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_merge_gathered_groups.argtypes = [vp, C.c_int32, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_ctx_sync.argtypes = [vp]
    L.lmx_merge_raw.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.lmx_renderer_params_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(RendererParams))]
    L.lmx_renderer_params_save.argtypes = [C.POINTER(RendererParams), C.c_char_p]
    L.lmx_renderer_params_free.argtypes = [C.POINTER(RendererParams)]
    L.lmx_renderer_params_free.restype = None
    L.lmx_cluster_matches.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(ClusterParams), vp, C.c_size_t,
                                      C.POINTER(C.c_size_t), vp, C.c_size_t]
    L.lmx_ctx_set_cluster_sidecar.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(ClusterParams)]
    L.lmx_ctx_collect_clusters.argtypes = [vp, C.c_int32, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t]
    L.lmx_ctx_debug_read.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_size_t]
    L.lmx_debug_orientation_labels.argtypes = [C.c_int32, vp, vp, C.c_size_t, vp]
    L.lmx_debug_introsort_perm.argtypes = [vp, vp, C.c_int32, vp]
    L.lmx_debug_introsort_perm_score.argtypes = [vp, C.c_int32, vp]
    L.lmx_debug_device_sort_perm.argtypes = [C.c_int32, vp, vp, C.c_int32, vp]
    L.lmx_ctx_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.lmx_kernel_name.argtypes = [C.c_int32]
    L.lmx_kernel_name.restype = C.c_char_p
    L.lmx_ctx_device_kernel_name.argtypes = [vp, C.c_int32]
    L.lmx_ctx_device_kernel_name.restype = C.c_char_p
    L.lmx_ctx_set_profiling.argtypes = [vp, C.c_int32]
    L.lmx_ctx_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.lmx_ctx_reset_profiling.argtypes = [vp]
    L.lmx_ctx_algorithmic_bytes.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    _lib = L
    return L


class LmxError(RuntimeError):
    """Raised for any non-OK lmx_status.  `.status` carries the code; LMX_ERR_SHAPE is the analogue of the
    cv::Exception upstream's CV_Asserts throw."""

    def __init__(self, status, message):
        super().__init__("lmx status %d: %s" % (status, message))
        self.status = status


def check(status):
    if status != LMX_OK:
        raise LmxError(status, lib().lmx_last_error().decode(errors="replace"))
