"""ctypes binding of libcloops_hip.so (C ABI: include/cloops_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or no MI355X is
visible, every entry point raises.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcloops_hip.so")

CL_OK = 0
CL_ERR_ARG = -1
CL_ERR_HIP = -2
CL_ERR_EMPTY = -3
CL_ERR_DOMAIN = -4
CL_ERR_GRID = -5
CL_ERR_NODEVICE = -6
CL_ERR_HASH = -7

VARIANT_CDBSCAN1 = 1
VARIANT_CDBSCAN2 = 2
VARIANT_BLOCK = 3

# every symbol include/cloops_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "cl_last_error", "cl_device_count", "cl_chrom_create", "cl_chrom_destroy", "cl_chrom_size",
    "cl_cluster", "cl_get_boxes", "cl_neighbor_counts", "cl_labels_device", "cl_set_profiling",
    "cl_get_timing", "cl_version", "cl_host_alloc", "cl_host_free", "cl_cluster_async", "cl_wait", "cl_boxes_host",
    "cl_dist_summary", "cl_dist_bin_hist", "cl_last_n_in", "cl_sig_counts", "cl_cluster_weighted",
    "cl_set_layout_reuse", "cl_set_sort_index", "cl_set_device_labels", "cl_set_table_export", "cl_cand_reset", "cl_cand_append", "cl_cand_finish", "cl_cluster_step_async", "cl_step_result",
    "cl_set_count_reuse", "cl_set_count_floor", "cl_set_count_thresholds", "cl_set_eps_list", "cl_chrom_set_stream", "cl_last_region_mode", "cl_debug_arena_overcommit", "cl_chrom_subsample", "cl_stream_create", "cl_stream_destroy", "cl_set_traversal", "cl_cand_finish_device", "cl_cluster_pairs_async", "cl_cluster_rowmask_async", "cl_last_n_labelled", "cl_set_pairs_defer", "cl_pairs_sync", "cl_sweep_plan", "cl_chrom_drop_indexes",
]


class ClBox(ctypes.Structure):
    _fields_ = [("min_x", ctypes.c_int32), ("max_x", ctypes.c_int32), ("min_y", ctypes.c_int32),
                ("max_y", ctypes.c_int32), ("count", ctypes.c_int32)]


class ClTiming(ctypes.Structure):
    _fields_ = [("ms_keys", ctypes.c_float), ("ms_sort", ctypes.c_float), ("ms_region", ctypes.c_float),
                ("ms_union", ctypes.c_float), ("ms_border", ctypes.c_float), ("ms_table", ctypes.c_float),
                ("ms_d2h", ctypes.c_float), ("ms_total", ctypes.c_float), ("n_in", ctypes.c_int64),
                ("n_strips", ctypes.c_int64), ("ms_bracket", ctypes.c_float), ("ms_band", ctypes.c_float),
                ("n_queried", ctypes.c_int64)]


DIST_LOGBINS = 3840


class ClDsummary(ctypes.Structure):
    _fields_ = [("n_all", ctypes.c_int64 * 2), ("n_pos", ctypes.c_int64 * 2), ("sumx", ctypes.c_double * 2),
                ("sumxx", ctypes.c_double * 2), ("xshift", ctypes.c_double), ("loghist", ctypes.c_uint64 * DIST_LOGBINS),
                ("fine_lo", ctypes.c_int64), ("fine", ctypes.c_uint64 * 2048)]


class CloopsHipError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libcloops_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the shared library (raises if it has not been built -- see cloops_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # (The library reads and writes no environment variable; how many runs execute side by side is decided by the streams the
    # application hands to cl_chrom_create -- the sweep driver shares three per GPU)
    path = SO_PATH
    if os.environ.get("CLOOPS_DEVEL_LIB") == "1":          # developer build with ablation knobs (cloops_amd/build.py --devel)
        path = SO_PATH.replace(".so", "_devel.so")
    if not os.path.exists(path):
        raise ImportError(
            "libcloops_hip.so is missing (%s). Build it with `python -m cloops_amd.build` "
            "(needs hipcc); there is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    i32p = ctypes.POINTER(ctypes.c_int32)
    vp = ctypes.c_void_p
    lib.cl_last_error.restype = ctypes.c_char_p
    lib.cl_last_error.argtypes = []
    lib.cl_device_count.restype = ctypes.c_int
    lib.cl_version.restype = ctypes.c_int
    lib.cl_chrom_create.restype = ctypes.c_int
    lib.cl_chrom_create.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(vp)]
    lib.cl_chrom_destroy.restype = None
    lib.cl_chrom_destroy.argtypes = [vp]
    lib.cl_chrom_size.restype = ctypes.c_int64
    lib.cl_chrom_size.argtypes = [vp]
    lib.cl_cluster.restype = ctypes.c_int
    lib.cl_cluster.argtypes = [vp, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, i32p, i32p]
    lib.cl_cluster_weighted.restype = ctypes.c_int
    lib.cl_cluster_weighted.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, i32p, i32p]
    lib.cl_cluster_async.restype = ctypes.c_int
    lib.cl_cluster_async.argtypes = [vp, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    lib.cl_wait.restype = ctypes.c_int
    lib.cl_wait.argtypes = [vp, i32p, i32p]
    lib.cl_boxes_host.restype = vp
    lib.cl_boxes_host.argtypes = [vp]
    lib.cl_dist_summary.restype = ctypes.c_int
    lib.cl_dist_summary.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ClDsummary)]
    lib.cl_dist_bin_hist.restype = ctypes.c_int
    lib.cl_dist_bin_hist.argtypes = [vp, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    lib.cl_set_device_labels.restype = None
    lib.cl_set_device_labels.argtypes = [vp, ctypes.c_int]
    lib.cl_set_table_export.restype = None
    lib.cl_set_table_export.argtypes = [vp, ctypes.c_int]
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.cl_cand_reset.restype = ctypes.c_int
    lib.cl_cand_reset.argtypes = [vp]
    lib.cl_cand_append.restype = ctypes.c_int
    lib.cl_cand_append.argtypes = [vp, ctypes.c_int32, i64p, i64p]
    lib.cl_cluster_step_async.restype = ctypes.c_int
    lib.cl_cluster_step_async.argtypes = [vp, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]
    lib.cl_step_result.restype = ctypes.c_int
    lib.cl_step_result.argtypes = [vp, i64p, i64p, ctypes.POINTER(ClDsummary)]
    lib.cl_cand_finish.restype = ctypes.c_int
    lib.cl_cand_finish.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int64, i64p]
    lib.cl_cluster_pairs_async.restype = ctypes.c_int
    lib.cl_cluster_pairs_async.argtypes = [vp, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64]
    lib.cl_cluster_rowmask_async.restype = ctypes.c_int
    lib.cl_cluster_rowmask_async.argtypes = [vp, ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64]
    lib.cl_set_pairs_defer.restype = None
    lib.cl_set_pairs_defer.argtypes = [vp, ctypes.c_int]
    lib.cl_pairs_sync.restype = ctypes.c_int
    lib.cl_pairs_sync.argtypes = [vp]
    lib.cl_last_n_labelled.restype = ctypes.c_int64
    lib.cl_last_n_labelled.argtypes = [vp]
    lib.cl_cand_finish_device.restype = ctypes.c_int
    lib.cl_cand_finish_device.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(vp), i64p]
    lib.cl_sig_counts.restype = ctypes.c_int
    lib.cl_sig_counts.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp, ctypes.POINTER(ctypes.c_int64)]
    lib.cl_last_n_in.restype = ctypes.c_int64
    lib.cl_last_n_in.argtypes = [vp]
    lib.cl_get_boxes.restype = ctypes.c_int
    lib.cl_get_boxes.argtypes = [vp, vp]
    lib.cl_neighbor_counts.restype = ctypes.c_int
    lib.cl_neighbor_counts.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp]
    lib.cl_labels_device.restype = vp
    lib.cl_labels_device.argtypes = [vp]
    lib.cl_set_profiling.restype = None
    lib.cl_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.cl_set_layout_reuse.restype = None
    lib.cl_set_layout_reuse.argtypes = [vp, ctypes.c_int]
    lib.cl_set_sort_index.restype = None
    lib.cl_set_sort_index.argtypes = [vp, ctypes.c_int]
    lib.cl_set_count_reuse.restype = None
    lib.cl_set_count_reuse.argtypes = [vp, ctypes.c_int]
    lib.cl_set_traversal.restype = None
    lib.cl_set_traversal.argtypes = [vp, ctypes.c_int]
    lib.cl_set_count_floor.restype = None
    lib.cl_set_count_floor.argtypes = [vp, ctypes.c_int32]
    lib.cl_chrom_set_stream.restype = ctypes.c_int
    lib.cl_chrom_set_stream.argtypes = [vp, vp]
    lib.cl_sweep_plan.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
    lib.cl_chrom_drop_indexes.argtypes = [vp]
    lib.cl_set_eps_list.restype = None
    lib.cl_set_eps_list.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
    lib.cl_set_count_thresholds.restype = None
    lib.cl_set_count_thresholds.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
    lib.cl_last_region_mode.restype = ctypes.c_int
    lib.cl_last_region_mode.argtypes = [vp]
    lib.cl_stream_create.restype = vp
    lib.cl_stream_create.argtypes = [ctypes.c_int]
    lib.cl_stream_destroy.restype = None
    lib.cl_stream_destroy.argtypes = [vp]
    lib.cl_chrom_subsample.restype = ctypes.c_int
    lib.cl_chrom_subsample.argtypes = [vp, vp, ctypes.c_int64, ctypes.POINTER(vp)]
    lib.cl_debug_arena_overcommit.restype = None
    lib.cl_debug_arena_overcommit.argtypes = [ctypes.c_int64]
    lib.cl_get_timing.restype = ctypes.c_int
    lib.cl_get_timing.argtypes = [vp, ctypes.POINTER(ClTiming)]
    lib.cl_host_alloc.restype = vp
    lib.cl_host_alloc.argtypes = [ctypes.c_int64]
    lib.cl_host_free.restype = None
    lib.cl_host_free.argtypes = [vp]
    _lib = lib
    return lib


def check(rc):
    if rc != CL_OK:
        msg = load().cl_last_error()
        raise CloopsHipError(rc, msg.decode() if msg else "")
