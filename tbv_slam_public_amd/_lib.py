"""ctypes binding of libcfear_hip.so (include/cfear_hip.h).

The library is built in-tree by tbv_slam_public_amd/csrc/Makefile (hipcc, gfx950).  There is no
CPU fallback: loading fails loudly if the .so is missing, and cfear_ctx_create fails with
CFEAR_ERR_NO_DEVICE when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libcfear_hip.so")

OK = 0
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_CAPACITY = -1, -2, -3
ERR_TOO_FEW_RESIDUALS, ERR_SOLVER, ERR_EMPTY_CLOUD, ERR_NO_DEVICE = -4, -5, -6, -7
ERR_IO, ERR_FORMAT = -8, -9

P2P, P2L, P2D = 0, 1, 2
LOSS = {"None": 0, "Huber": 1, "Cauchy": 2, "SoftLOne": 3, "Combined": 4, "Tukey": 5}
COST = {"P2P": 0, "P2L": 1, "P2D": 2}


class CfearError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("cfear status %d: %s" % (status, msg))
        self.status = status


class PolarDesc(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("stride", C.c_int32), ("batch", C.c_int32),
                ("batch_stride", C.c_int64)]


class KStrongParams(C.Structure):
    _fields_ = [("k_strongest", C.c_int32), ("z_min", C.c_float), ("range_res", C.c_float),
                ("min_distance", C.c_float), ("want_peaks", C.c_int32)]


class KStrongOut(C.Structure):
    _fields_ = [("sel_range", C.c_void_p), ("sel_intensity", C.c_void_p), ("sel_count", C.c_void_p),
                ("is_peak", C.c_void_p), ("xyzi", C.c_void_p), ("n_points", C.c_void_p),
                ("xyzi_peaks", C.c_void_p), ("n_peaks", C.c_void_p)]


class CacfarParams(C.Structure):
    _fields_ = [("window_size", C.c_int32), ("nb_guard_cells", C.c_int32), ("false_alarm_rate", C.c_float),
                ("range_res", C.c_float), ("z_min", C.c_float), ("min_distance", C.c_float),
                ("max_distance", C.c_double)]


class Cell(C.Structure):
    _fields_ = [("mean", C.c_double * 2), ("normal", C.c_double * 2), ("cov", C.c_double * 4),
                ("scale", C.c_double), ("avg_intensity", C.c_double), ("lambda_min", C.c_double),
                ("lambda_max", C.c_double), ("nsamples", C.c_int32), ("pad", C.c_int32)]


CELL_DTYPE = np.dtype([("mean", "<f8", (2,)), ("normal", "<f8", (2,)), ("cov", "<f8", (4,)),
                       ("scale", "<f8"), ("avg_intensity", "<f8"), ("lambda_min", "<f8"),
                       ("lambda_max", "<f8"), ("nsamples", "<i4"), ("pad", "<i4")])
assert CELL_DTYPE.itemsize == C.sizeof(Cell) == 104


class FeatureParams(C.Structure):
    _fields_ = [("radius", C.c_float), ("downsample_factor", C.c_double), ("origin", C.c_double * 2),
                ("weight_intensity", C.c_int32), ("compensate", C.c_int32), ("mot", C.c_double * 3),
                ("ccw", C.c_int32), ("pad", C.c_int32)]


class RegParams(C.Structure):
    _fields_ = [("cost", C.c_int32), ("loss", C.c_int32), ("loss_limit", C.c_double),
                ("weight_opt", C.c_int32), ("max_itr_association", C.c_int32),
                ("max_itr_solver", C.c_int32), ("min_itr", C.c_int32), ("radius", C.c_double),
                ("cov_scale", C.c_double), ("regularization", C.c_double),
                ("score_tolerance", C.c_double), ("itr", C.c_int32), ("pad", C.c_int32)]


class RegResult(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("score", C.c_double), ("final_cost", C.c_double),
                ("num_residuals", C.c_int32), ("outer_iters", C.c_int32), ("lm_iters", C.c_int32),
                ("status", C.c_int32), ("last_relative_decrease", C.c_double), ("reserved", C.c_double)]


RESULT_DTYPE = np.dtype([("pose", "<f8", (3,)), ("score", "<f8"), ("final_cost", "<f8"),
                         ("num_residuals", "<i4"), ("outer_iters", "<i4"), ("lm_iters", "<i4"),
                         ("status", "<i4"), ("last_relative_decrease", "<f8"), ("reserved", "<f8")])
assert RESULT_DTYPE.itemsize == C.sizeof(RegResult) == 72


class RegJob(C.Structure):
    _fields_ = [("scans", C.POINTER(C.c_void_p)), ("n_scans", C.c_int32), ("pad", C.c_int32),
                ("poses_xyt", C.POINTER(C.c_double))]


CANDIDATE_DTYPE = np.dtype([("target", "<i4"), ("source", "<i4"), ("target_xyt", "<f8", (3,)), ("source_xyt", "<f8", (3,))])
assert CANDIDATE_DTYPE.itemsize == 56


class CovSamplingParams(C.Structure):
    _fields_ = [("xy_range", C.c_double), ("yaw_range", C.c_double), ("samples_per_axis", C.c_int32),
                ("pad", C.c_int32), ("covariance_scaler", C.c_double)]


class CoralParams(C.Structure):
    _fields_ = [("radius", C.c_double), ("weight_res_intensity", C.c_int32), ("pad", C.c_int32)]


class CoralJob(C.Structure):
    _fields_ = [("ref_xyzi", C.c_void_p), ("src_xyzi", C.c_void_p), ("n_ref", C.c_int32), ("n_src", C.c_int32),
                ("ref_pose", C.c_double * 3), ("src_pose", C.c_double * 3), ("offset", C.c_double * 3)]


class CoralResult(C.Structure):
    _fields_ = [("joint", C.c_double), ("sep", C.c_double), ("overlap", C.c_double), ("valid", C.c_int32),
                ("count_valid", C.c_int32), ("status", C.c_int32), ("pad", C.c_int32)]


CORAL_RESULT_DTYPE = np.dtype([("joint", "<f8"), ("sep", "<f8"), ("overlap", "<f8"), ("valid", "<i4"),
                               ("count_valid", "<i4"), ("status", "<i4"), ("pad", "<i4")])


class ScParams(C.Structure):
    _fields_ = [("num_ring", C.c_int32), ("num_sector", C.c_int32), ("max_radius", C.c_double),
                ("search_ratio", C.c_double), ("desc_function", C.c_int32), ("pad", C.c_int32),
                ("desc_divider", C.c_double), ("no_point", C.c_double)]


class ScCloud(C.Structure):
    _fields_ = [("xyzi", C.c_void_p), ("n", C.c_int32), ("pad", C.c_int32)]


class ScManagerParams(C.Structure):
    _fields_ = [("sc", ScParams), ("num_candidates_from_tree", C.c_int32), ("n_candidates", C.c_int32),
                ("odom_sigma_error", C.c_double), ("odometry_coupled_closure", C.c_int32), ("augment_sc", C.c_int32),
                ("distance_exclude_recent", C.c_double), ("pad", C.c_int64)]


SC_CANDIDATE_DTYPE = np.dtype([("min_dist", "<f8"), ("min_dist_sc", "<f8"), ("min_dist_odom", "<f8"),
                               ("yaw_diff_rad", "<f4"), ("nn_idx", "<i4"), ("argmin_shift", "<i4"), ("pad", "<i4"),
                               ("Taug", "<f8", (3,))])
assert SC_CANDIDATE_DTYPE.itemsize == 64


class VerifyParams(C.Structure):
    _fields_ = [("align_intercept", C.c_double), ("align_coef", C.c_double * 6), ("loop_intercept", C.c_double),
                ("loop_coef", C.c_double * 3), ("model_threshold", C.c_double), ("all_candidates", C.c_int32),
                ("verification_disabled", C.c_int32), ("use_covariance_sampling", C.c_int32), ("pad", C.c_int32),
                ("coral", CoralParams), ("sampling", CovSamplingParams)]


class VerifyJob(C.Structure):
    _fields_ = [("from_scan", C.c_void_p), ("to_scan", C.c_void_p), ("from_peaks", C.c_void_p),
                ("to_peaks", C.c_void_p), ("n_from", C.c_int32), ("n_to", C.c_int32), ("from_pose", C.c_double * 3),
                ("t_be_guess", C.c_double * 3), ("sc_sim", C.c_double), ("odom_bounds", C.c_double),
                ("group", C.c_int32), ("pad", C.c_int32)]


class VerifyResult(C.Structure):
    _fields_ = [("t_be", C.c_double * 3), ("cov", C.c_double * 36), ("coral", C.c_double * 3),
                ("cfear", C.c_double * 3), ("alignment_quality", C.c_double), ("odom_bounds", C.c_double),
                ("sc_sim", C.c_double), ("probability", C.c_double), ("reg_ok", C.c_int32),
                ("cov_sampled", C.c_int32), ("accepted", C.c_int32), ("rank", C.c_int32), ("reg", RegResult)]


VERIFY_RESULT_DTYPE = np.dtype([("t_be", "<f8", (3,)), ("cov", "<f8", (6, 6)), ("coral", "<f8", (3,)),
                                ("cfear", "<f8", (3,)), ("alignment_quality", "<f8"), ("odom_bounds", "<f8"),
                                ("sc_sim", "<f8"), ("probability", "<f8"), ("reg_ok", "<i4"), ("cov_sampled", "<i4"),
                                ("accepted", "<i4"), ("rank", "<i4"), ("reg", RESULT_DTYPE)])
assert VERIFY_RESULT_DTYPE.itemsize == C.sizeof(VerifyResult) == 480


class OdometryParams(C.Structure):
    _fields_ = [("filter_type", C.c_int32), ("kstrong", KStrongParams), ("cacfar", CacfarParams),
                ("reg", RegParams), ("res", C.c_float), ("submap_scan_size", C.c_int32),
                ("weight_intensity", C.c_int32), ("use_guess", C.c_int32), ("compensate", C.c_int32),
                ("radar_ccw", C.c_int32), ("use_keyframe", C.c_int32), ("rotate_ccw", C.c_int32),
                ("min_keyframe_dist", C.c_double), ("min_keyframe_rot_deg", C.c_double),
                ("downsample_factor", C.c_double), ("estimate_cov_by_sampling", C.c_int32), ("keep_nodes", C.c_int32),
                ("cov_sampling", CovSamplingParams)]


class FrameInfo(C.Structure):
    _fields_ = [("pose", C.c_double * 3), ("n_points", C.c_int32), ("n_cells", C.c_int32),
                ("keyframe_added", C.c_int32), ("reg_status", C.c_int32), ("outer_iters", C.c_int32),
                ("lm_iters", C.c_int32), ("score", C.c_double)]


FRAMEINFO_DTYPE = np.dtype([("pose", "<f8", (3,)), ("n_points", "<i4"), ("n_cells", "<i4"),
                            ("keyframe_added", "<i4"), ("reg_status", "<i4"), ("outer_iters", "<i4"),
                            ("lm_iters", "<i4"), ("score", "<f8")])
assert FRAMEINFO_DTYPE.itemsize == C.sizeof(FrameInfo) == 56

# every symbol include/cfear_hip.h declares
EXPORTS = [
    "cfear_abi_version", "cfear_status_string", "cfear_ctx_create", "cfear_ctx_destroy",
    "cfear_ctx_synchronize", "cfear_last_error", "cfear_ctx_profile_enable", "cfear_ctx_profile_read",
    "cfear_filter_kstrongest", "cfear_filter_cacfar", "cfear_compensate", "cfear_scan_create",
    "cfear_scan_from_cells", "cfear_scan_size", "cfear_scan_get_cells", "cfear_scan_destroy",
    "cfear_reg_params_default", "cfear_register", "cfear_register_batch", "cfear_get_cost",
    "cfear_get_cost_batch", "cfear_cov_sampling_params_default", "cfear_covariance_by_sampling",
    "cfear_covariance_by_sampling_batch", "cfear_coral_params_default", "cfear_coral_quality",
    "cfear_coral_quality_batch", "cfear_sc_params_default", "cfear_sc_descriptors", "cfear_sc_distance_batch",
    "cfear_polar_rotate_ccw", "cfear_scan_closest_idx",
    "cfear_sc_manager_params_default", "cfear_sc_manager_create", "cfear_sc_manager_add", "cfear_sc_manager_detect",
    "cfear_sc_manager_size", "cfear_sc_manager_destroy", "cfear_verify_params_default", "cfear_verify_loop_candidates", "cfear_verify_by_odometry", "cfear_verify_apply_constraints",
    "cfear_cost_prepare", "cfear_cost_num_blocks", "cfear_cost_num_residuals", "cfear_cost_get_blocks",
    "cfear_cost_evaluate", "cfear_cost_normal_eq", "cfear_cost_destroy",
    "cfear_odometry_params_default", "cfear_odometry_params_preset", "cfear_odometry_create", "cfear_odometry_process",
    "cfear_odometry_process_prefetch", "cfear_odometry_get_covariance", "cfear_odometry_destroy",
    "cfear_odometry_get_scan", "cfear_odometry_get_cloud", "cfear_odometry_get_peaks", "cfear_odometry_process_clouds",
    "cfear_odometry_process_offsets", "cfear_odometry_discard_prefetch",
    "cfear_keyframe_based_fuse", "cfear_acc_vel_sanity_check", "cfear_filter_kstrongest_legacy", "cfear_filter_kstrongest_rowkeys",
    "cfear_graph_save", "cfear_graph_load", "cfear_graph_size", "cfear_graph_node_at", "cfear_graph_destroy",
    "cfear_pose3d_from_xyt", "cfear_pose3d_to_xyt", "cfear_odometry_get_constraint",
    "cfear_shard_range", "cfear_gather_records", "cfear_register_batch_sharded", "cfear_verify_loop_candidates_sharded",
    "cfear_rccl_allgather", "cfear_rccl_allgather_device", "cfear_pgo_params_default", "cfear_pgo_solve",
    "cfear_scan_table_create", "cfear_scan_table_size", "cfear_scan_table_destroy", "cfear_register_candidates",
    "cfear_ctx_get_stream", "cfear_ctx_set_option", "cfear_ctx_get_option",
    "cfear_rccl_unique_id", "cfear_rccl_comm_init", "cfear_rccl_comm_destroy",
    "cfear_candidate_pipe_create", "cfear_candidate_pipe_submit", "cfear_candidate_pipe_collect", "cfear_candidate_pipe_destroy", "cfear_candidate_pipe_stats",
]

PIPE_GRAPH, PIPE_TIMING = 1, 2      # enum { CFEAR_PIPE_GRAPH, CFEAR_PIPE_TIMING }


class RcclComm(C.Structure):        # cfear_rccl_comm
    _fields_ = [("ctx", C.c_void_p), ("nccl_comm", C.c_void_p), ("world", C.c_int32), ("pad", C.c_int32)]


# enum cfear_option (include/cfear_hip.h): test / measurement hooks of a context
OPT_FUSED_DECODE, OPT_MATCHER_LDS_KB, OPT_MATCHER_WAVES, OPT_HOST_TIMELINE, OPT_COUNT = 0, 1, 2, 3, 4


class PgoParams(C.Structure):
    _fields_ = [("loop_vxx", C.c_double), ("loop_vyy", C.c_double), ("loop_vtt", C.c_double), ("odom_vxx", C.c_double),
                ("odom_vyy", C.c_double), ("odom_vtt", C.c_double), ("loop_scaling", C.c_double),
                ("replace_cov_by_identity", C.c_int32), ("max_num_iterations", C.c_int32), ("loop_loss_limit", C.c_double)]


class PgoSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32), ("usable", C.c_int32),
                ("num_residual_blocks", C.c_int32), ("linear_iterations", C.c_int32)]



class Pose3d(C.Structure):
    _fields_ = [("p", C.c_double * 3), ("q", C.c_double * 4)]


class GraphCloud(C.Structure):
    _fields_ = [("xyzi", C.c_void_p), ("n", C.c_int32), ("seq", C.c_uint32), ("stamp", C.c_uint64), ("frame_id", C.c_char_p)]


class GraphConstraint(C.Structure):
    _fields_ = [("id_begin", C.c_uint64), ("id_end", C.c_uint64), ("t_be", Pose3d), ("information", C.c_double * 36),
                ("type", C.c_int32), ("n_quality", C.c_int32), ("quality_keys", C.POINTER(C.c_char_p)),
                ("quality_values", C.POINTER(C.c_double)), ("info", C.c_char_p)]


class GraphNode(C.Structure):
    _fields_ = [("T", Pose3d), ("Tgt", Pose3d), ("has_Tgt", C.c_int32), ("idx", C.c_uint32), ("stamp", C.c_uint64),
                ("motion", C.c_double * 16), ("cloud_peaks", GraphCloud), ("cloud_nopeaks", GraphCloud),
                ("has_normal", C.c_int32), ("input_is_nopeaks", C.c_int32), ("normal_input", GraphCloud),
                ("cells", C.c_void_p), ("n_cells", C.c_int32), ("radius", C.c_float), ("weight_intensity", C.c_int32),
                ("pad", C.c_int32), ("constraints", C.POINTER(GraphConstraint)), ("n_constraints", C.c_int32), ("pad2", C.c_int32)]


_LIB = None


def lib():
    """Loads libcfear_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise ImportError("libcfear_hip.so not built: run `make -C %s/csrc` (or __graft_entry__.build())" % _HERE)
    # PyTorch-ROCm bundles its own libamdhip64; it must be the process's HIP runtime BEFORE this
    # library is loaded, otherwise two runtimes coexist and torch device pointers / streams cannot be
    # shared with the kernels (torch is plumbing here: device memory, streams, torch.distributed).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    vp = C.c_void_p
    L.cfear_abi_version.restype = C.c_int
    L.cfear_status_string.restype = C.c_char_p
    L.cfear_status_string.argtypes = [C.c_int]
    L.cfear_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.cfear_ctx_destroy.argtypes = [vp]
    L.cfear_ctx_synchronize.argtypes = [vp]
    L.cfear_ctx_get_stream.argtypes = [vp, C.POINTER(C.c_void_p)]
    L.cfear_ctx_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    L.cfear_ctx_get_option.argtypes = [vp, C.c_int32, C.POINTER(C.c_int64)]
    L.cfear_last_error.argtypes = [vp]
    L.cfear_last_error.restype = C.c_char_p
    L.cfear_ctx_profile_enable.argtypes = [vp, C.c_int]
    L.cfear_ctx_profile_read.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64), C.c_int, C.c_int]
    L.cfear_scan_table_create.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(vp)]
    L.cfear_scan_table_size.argtypes = [vp]
    L.cfear_scan_table_destroy.argtypes = [vp]
    L.cfear_register_candidates.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
    L.cfear_filter_kstrongest.argtypes = [vp, vp, C.POINTER(PolarDesc), C.POINTER(KStrongParams),
                                          C.POINTER(KStrongOut)]
    L.cfear_filter_cacfar.argtypes = [vp, vp, C.POINTER(PolarDesc), C.POINTER(CacfarParams), vp, vp,
                                      C.c_int32, vp]
    L.cfear_compensate.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_double), C.c_int32]
    L.cfear_scan_create.argtypes = [vp, vp, C.c_int32, C.POINTER(FeatureParams), C.POINTER(vp)]
    L.cfear_scan_from_cells.argtypes = [vp, vp, C.c_int32, C.POINTER(vp)]
    L.cfear_scan_size.argtypes = [vp]
    L.cfear_scan_get_cells.argtypes = [vp, vp, C.c_int32]
    L.cfear_scan_destroy.argtypes = [vp]
    L.cfear_reg_params_default.argtypes = [C.POINTER(RegParams)]
    L.cfear_reg_params_default.restype = None
    L.cfear_register.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(C.c_double),
                                 C.POINTER(RegParams), C.POINTER(RegResult)]
    L.cfear_register_batch.argtypes = [vp, C.POINTER(RegJob), C.c_int32, C.POINTER(RegParams), vp]
    L.cfear_get_cost_batch.argtypes = [vp, C.POINTER(RegJob), C.c_int32, C.POINTER(RegParams), vp]
    L.cfear_cov_sampling_params_default.argtypes = [C.POINTER(CovSamplingParams)]
    L.cfear_cov_sampling_params_default.restype = None
    L.cfear_covariance_by_sampling.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(C.c_double),
                                               C.POINTER(RegParams), C.POINTER(RegResult),
                                               C.POINTER(CovSamplingParams), C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.cfear_covariance_by_sampling_batch.argtypes = [vp, C.POINTER(RegJob), C.c_int32, C.POINTER(RegParams), vp,
                                                     C.POINTER(CovSamplingParams), vp, vp, vp]
    L.cfear_coral_params_default.argtypes = [C.POINTER(CoralParams)]
    L.cfear_coral_params_default.restype = None
    L.cfear_coral_quality.argtypes = [vp, C.POINTER(CoralJob), C.POINTER(CoralParams), vp, vp]
    L.cfear_coral_quality_batch.argtypes = [vp, C.POINTER(CoralJob), C.c_int32, C.POINTER(CoralParams), vp, vp]
    L.cfear_sc_params_default.argtypes = [C.POINTER(ScParams)]
    L.cfear_sc_params_default.restype = None
    L.cfear_sc_descriptors.argtypes = [vp, C.POINTER(ScCloud), C.c_int32, C.POINTER(ScParams), C.POINTER(C.c_double),
                                       C.c_int32, vp, vp, vp]
    L.cfear_sc_distance_batch.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, C.POINTER(ScParams), vp, vp]
    L.cfear_sc_manager_params_default.argtypes = [C.POINTER(ScManagerParams)]
    L.cfear_sc_manager_params_default.restype = None
    L.cfear_sc_manager_create.argtypes = [vp, C.POINTER(ScManagerParams), C.POINTER(vp)]
    L.cfear_sc_manager_add.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_double)]
    L.cfear_sc_manager_detect.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.cfear_sc_manager_size.argtypes = [vp]
    L.cfear_sc_manager_destroy.argtypes = [vp]
    L.cfear_scan_closest_idx.argtypes = [vp, vp, C.c_int32, C.c_double, vp]
    L.cfear_polar_rotate_ccw.argtypes = [vp, vp, C.POINTER(PolarDesc), vp, C.c_int32, C.c_int64]
    L.cfear_verify_params_default.argtypes = [C.POINTER(VerifyParams)]
    L.cfear_verify_params_default.restype = None
    L.cfear_verify_loop_candidates.argtypes = [vp, C.POINTER(VerifyJob), C.c_int32, C.POINTER(VerifyParams), vp]
    L.cfear_verify_apply_constraints.argtypes = [vp, C.c_int32, C.POINTER(VerifyParams), vp]
    L.cfear_verify_by_odometry.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_double)]
    L.cfear_get_cost.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(C.c_double), C.POINTER(RegParams),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.cfear_cost_prepare.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(C.c_double),
                                     C.POINTER(RegParams), C.c_int32, C.POINTER(vp)]
    L.cfear_cost_num_blocks.argtypes = [vp]
    L.cfear_cost_num_residuals.argtypes = [vp]
    L.cfear_cost_get_blocks.argtypes = [vp, vp, vp]
    L.cfear_cost_evaluate.argtypes = [vp, C.POINTER(C.c_double), vp, vp]
    L.cfear_cost_normal_eq.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cfear_cost_destroy.argtypes = [vp]
    L.cfear_odometry_params_default.argtypes = [C.POINTER(OdometryParams)]
    L.cfear_odometry_params_default.restype = None
    L.cfear_odometry_process_clouds.argtypes = [vp, C.POINTER(ScCloud), C.POINTER(ScCloud), vp]
    L.cfear_odometry_get_scan.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.cfear_odometry_get_cloud.argtypes = [vp, C.c_int32, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.cfear_odometry_get_peaks.argtypes = [vp, C.c_int32, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.cfear_odometry_params_preset.argtypes = [C.POINTER(OdometryParams), C.c_int32, C.c_int32]
    L.cfear_odometry_create.argtypes = [vp, C.c_int32, C.POINTER(PolarDesc), C.POINTER(OdometryParams),
                                        C.POINTER(vp)]
    L.cfear_odometry_process.argtypes = [vp, vp, vp]
    L.cfear_odometry_process_prefetch.argtypes = [vp, vp, vp, vp]
    L.cfear_odometry_process_offsets.argtypes = [vp, vp, vp, vp, vp]
    L.cfear_odometry_discard_prefetch.argtypes = [vp]
    L.cfear_filter_kstrongest_legacy.argtypes = [vp, vp, C.POINTER(PolarDesc), C.c_int32, C.c_double, C.c_double, C.c_double, vp, vp,
                                                 C.c_int32]
    L.cfear_filter_kstrongest_rowkeys.argtypes = [vp, vp, C.POINTER(PolarDesc), C.POINTER(KStrongParams), C.c_int32, vp, vp]
    L.cfear_pgo_params_default.argtypes = [C.POINTER(PgoParams)]
    L.cfear_pgo_params_default.restype = None
    L.cfear_pgo_solve.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.POINTER(PgoParams), C.POINTER(PgoSummary)]
    L.cfear_shard_range.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.cfear_gather_records.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    L.cfear_graph_save.argtypes = [C.c_char_p, C.POINTER(GraphNode), C.c_int32]
    L.cfear_graph_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.cfear_graph_size.argtypes = [vp]
    L.cfear_graph_node_at.argtypes = [vp, C.c_int32, C.POINTER(GraphNode)]
    L.cfear_graph_destroy.argtypes = [vp]
    L.cfear_pose3d_from_xyt.argtypes = [C.POINTER(C.c_double), C.POINTER(Pose3d)]
    L.cfear_pose3d_from_xyt.restype = None
    L.cfear_pose3d_to_xyt.argtypes = [C.POINTER(Pose3d), C.POINTER(C.c_double)]
    L.cfear_pose3d_to_xyt.restype = None
    L.cfear_odometry_get_constraint.argtypes = [vp, C.c_int32, C.POINTER(GraphConstraint)]
    L.cfear_keyframe_based_fuse.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_double, C.c_double]
    L.cfear_acc_vel_sanity_check.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.cfear_odometry_get_covariance.argtypes = [vp, vp, vp]
    L.cfear_odometry_destroy.argtypes = [vp]
    L.cfear_rccl_unique_id.argtypes = [vp]
    L.cfear_rccl_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(RcclComm)]
    L.cfear_rccl_comm_destroy.argtypes = [C.POINTER(RcclComm)]
    L.cfear_candidate_pipe_create.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(RcclComm), C.c_int32, C.c_int32,
                                              C.POINTER(vp)]
    L.cfear_candidate_pipe_submit.argtypes = [vp, vp, C.c_int32, C.POINTER(RegParams), C.POINTER(C.c_int64)]
    L.cfear_candidate_pipe_collect.argtypes = [vp, C.c_int64, vp]
    L.cfear_candidate_pipe_destroy.argtypes = [vp]
    L.cfear_candidate_pipe_stats.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    _LIB = L
    return L
