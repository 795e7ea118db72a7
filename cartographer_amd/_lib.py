"""ctypes loader of libcartographer_mi355x.so (the C ABI in include/cartographer_mi355x.h).

The library is the product: there is no Python or CPU fallback.  If the
shared object is missing, or no HIP device is usable, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CMX_SO_PATH") or os.path.join(_HERE, "lib", "libcartographer_mi355x.so")

OK, INVALID_ARGUMENT, DEVICE_ERROR, OUT_OF_MEMORY, UNSUPPORTED = range(5)


class CmxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{message} (status {status})")
        self.status = status


class Pose2d(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("theta", C.c_double)]


class Pose3d(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("q", C.c_double * 4)]


class Grid2DLimits(C.Structure):
    _fields_ = [("resolution", C.c_double), ("max_x", C.c_double), ("max_y", C.c_double),
                ("num_x_cells", C.c_int32), ("num_y_cells", C.c_int32),
                ("min_correspondence_cost", C.c_float), ("max_correspondence_cost", C.c_float)]


class RtOptions(C.Structure):
    _fields_ = [("linear_search_window", C.c_double), ("angular_search_window", C.c_double),
                ("translation_delta_cost_weight", C.c_double),
                ("rotation_delta_cost_weight", C.c_double)]


class Fast2DOptions(C.Structure):
    _fields_ = [("linear_search_window", C.c_double), ("angular_search_window", C.c_double),
                ("branch_and_bound_depth", C.c_int32)]


class Fast3DOptions(C.Structure):
    _fields_ = [("branch_and_bound_depth", C.c_int32), ("full_resolution_depth", C.c_int32),
                ("min_rotational_score", C.c_double), ("min_low_resolution_score", C.c_double),
                ("linear_xy_search_window", C.c_double), ("linear_z_search_window", C.c_double),
                ("angular_search_window", C.c_double)]


class MatchStats(C.Structure):
    _fields_ = [("candidates_scored", C.c_int64), ("coarse_candidates", C.c_int64),
                ("nodes_expanded", C.c_int64), ("num_scans", C.c_int32), ("expansion_launches", C.c_int32),
                ("device_ms", C.c_double), ("dominant_kernel_ms", C.c_double),
                ("expansion_ms", C.c_double), ("expansion_nodes", C.c_int64),
                ("expansion_lookups", C.c_int64), ("refined_candidates", C.c_int64),
                ("finalists", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Ceres2DOptions(C.Structure):
    _fields_ = [("occupied_space_weight", C.c_double), ("translation_weight", C.c_double),
                ("rotation_weight", C.c_double), ("use_nonmonotonic_steps", C.c_int32),
                ("max_num_iterations", C.c_int32)]


class CeresSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("termination", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class Ceres3DOptions(C.Structure):
    _fields_ = [("occupied_space_weight", C.c_double * 3), ("translation_weight", C.c_double),
                ("rotation_weight", C.c_double), ("num_pairs", C.c_int32),
                ("only_optimize_yaw", C.c_int32), ("use_nonmonotonic_steps", C.c_int32),
                ("max_num_iterations", C.c_int32)]


class IntensityVoxel(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32), ("count", C.c_int32),
                ("sum", C.c_float)]


INTENSITY_VOXEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32),
                                  ("count", np.int32), ("sum", np.float32)])


class Ceres3DPair(C.Structure):
    _fields_ = [("point_cloud_xyz", C.c_void_p), ("num_points", C.c_int32),
                ("resolution", C.c_float), ("voxels", C.c_void_p), ("num_voxels", C.c_int64),
                ("intensities", C.c_void_p), ("intensity_voxels", C.c_void_p),
                ("num_intensity_voxels", C.c_int64), ("intensity_weight", C.c_double),
                ("intensity_huber_scale", C.c_double), ("intensity_threshold", C.c_float),
                ("reserved", C.c_int32)]


class Candidate2D(C.Structure):
    """cmx_candidate2d = Candidate2D (SM2/correlative_scan_matcher_2d.h:69-98)."""
    _fields_ = [("scan_index", C.c_int32), ("x_index_offset", C.c_int32),
                ("y_index_offset", C.c_int32), ("score", C.c_float), ("x", C.c_double),
                ("y", C.c_double), ("orientation", C.c_double)]


class Voxel(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32), ("value", C.c_uint16),
                ("pad", C.c_uint16)]


VOXEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32), ("value", np.uint16),
                        ("pad", np.uint16)])


class NodeData3D(C.Structure):
    _fields_ = [("gravity_alignment", C.c_double * 4),
                ("high_resolution_point_cloud", C.c_void_p),
                ("num_high_resolution_points", C.c_int32),
                ("low_resolution_point_cloud", C.c_void_p),
                ("num_low_resolution_points", C.c_int32),
                ("rotational_scan_matcher_histogram", C.c_void_p),
                ("histogram_size", C.c_int32)]


class Ceres3DIntensityTerm(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("intensities", C.c_void_p), ("weight", C.c_double),
                ("huber_scale", C.c_double), ("intensity_threshold", C.c_float),
                ("reserved", C.c_int32)]


class Result3D(C.Structure):
    _fields_ = [("score", C.c_float), ("pose_estimate", Pose3d), ("rotational_score", C.c_float),
                ("low_resolution_score", C.c_float)]


# Every symbol include/cartographer_mi355x.h declares.
EXPORTED_SYMBOLS = [
    "cmx_version", "cmx_status_string", "cmx_last_error", "cmx_device_count", "cmx_set_stream",
    "cmx_rt2d_match", "cmx_rt2d_match_tsdf", "cmx_grid2d_create", "cmx_grid2d_destroy",
    "cmx_grid2d_get_limits", "cmx_grid2d_download", "cmx_grid2d_crop", "cmx_grid2d_insert",
    "cmx_rt2d_match_grid",
    "cmx_rt2d_match_grid_batch", "cmx_rt2d_match_grid_batch_resident",
    "cmx_grid3d_create", "cmx_grid3d_destroy", "cmx_grid3d_insert", "cmx_grid3d_info",
    "cmx_grid3d_download",
    "cmx_fast2d_create", "cmx_fast2d_create_from_grid", "cmx_fast2d_destroy", "cmx_fast2d_match",
    "cmx_fast2d_match_full_submap", "cmx_fast2d_match_batch",
    "cmx_fast2d_match_full_submap_batch", "cmx_cloud_upload",
    "cmx_cloud_destroy", "cmx_fast2d_match_full_submap_batch_resident", "cmx_fast2d_level_dims",
    "cmx_fast2d_level_cells", "cmx_fast2d_debug_prepare", "cmx_rt3d_match", "cmx_fast3d_create",
    "cmx_fast3d_destroy", "cmx_fast3d_match", "cmx_fast3d_match_full_submap",
    "cmx_fast3d_match_batch",
    "cmx_fast3d_level_info", "cmx_fast3d_level_cells",
    "cmx_ceres2d_match", "cmx_ceres2d_match_grid", "cmx_fast2d_refine_batch", "cmx_ceres3d_match",
    "cmx_ceres3d_match_grids", "cmx_rt2d_score_candidates", "cmx_rt3d_match_grid",
    "cmx_fast3d_refine_batch",
    "cmx_comm_init", "cmx_comm_destroy", "cmx_comm_num_devices", "cmx_comm_device_of",
    "cmx_comm_uses_rccl", "cmx_sizeof_match_stats",
    "cmx_fast2d_match_sharded", "cmx_fast3d_match_sharded", "cmx_shard_range",
    "cmx_pack_best_key", "cmx_unpack_best_key",
    "cmx_voxel_filter", "cmx_voxel_filter_indices", "cmx_adaptive_voxel_filter",
    "cmx_adaptive_voxel_filter_indices",
    "cmx_compute_histogram",
    "cmx_intensity_grid3d_create", "cmx_intensity_grid3d_destroy",
    "cmx_grid3d_insert_with_intensities", "cmx_intensity_grid3d_download",
    "cmx_ceres3d_match_grids_intensity",
]

# include/cartographer_mi355x_debug.h (test and tool switches, not part of the boundary).
DEBUG_SYMBOLS = ["cmx_debug_set", "cmx_debug_reset"]

_lib = None


def lib():
    """Loads the shared library (importing torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build it with `python -m cartographer_amd.build` "
            "(there is no fallback path)")
    # torch bundles libamdhip64.so.7; load it first so one runtime serves both.  A process that
    # never touches torch (the timing probes under tools/) may skip its minute of first import.
    if os.environ.get("CMX_SKIP_TORCH_IMPORT") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing only
            pass
    L = C.CDLL(SO_PATH)
    L.cmx_version.restype = C.c_char_p
    L.cmx_status_string.restype = C.c_char_p
    L.cmx_status_string.argtypes = [C.c_int]
    L.cmx_last_error.restype = C.c_char_p
    L.cmx_device_count.restype = C.c_int32
    L.cmx_sizeof_match_stats.restype = C.c_int32
    # (the struct has no size field: a mirror of another size would be overrun by the library)
    if L.cmx_sizeof_match_stats() != C.sizeof(MatchStats):
        raise CmxError(INVALID_ARGUMENT, "cmx_match_stats of the library is "
                       f"{L.cmx_sizeof_match_stats()} bytes, this module's mirror {C.sizeof(MatchStats)}")
    L.cmx_set_stream.argtypes = [C.c_int32, C.c_void_p]
    P = C.POINTER
    L.cmx_rt2d_match.argtypes = [P(RtOptions), P(Grid2DLimits), C.c_void_p, P(Pose2d), C.c_void_p,
                                 C.c_int32, C.c_int32, P(C.c_double), P(Pose2d), P(MatchStats)]
    L.cmx_rt2d_match_tsdf.argtypes = [P(RtOptions), P(Grid2DLimits), C.c_void_p, C.c_void_p,
                                      C.c_float, C.c_float, P(Pose2d), C.c_void_p, C.c_int32,
                                      C.c_int32, P(C.c_double), P(Pose2d), P(MatchStats)]
    L.cmx_grid2d_create.argtypes = [P(Grid2DLimits), C.c_void_p, C.c_int32, P(C.c_void_p)]
    L.cmx_grid2d_destroy.argtypes = [C.c_void_p]
    L.cmx_grid2d_destroy.restype = None
    L.cmx_grid2d_get_limits.argtypes = [C.c_void_p, P(Grid2DLimits)]
    L.cmx_grid2d_download.argtypes = [C.c_void_p, C.c_void_p]
    L.cmx_grid2d_crop.argtypes = [C.c_void_p]
    L.cmx_grid3d_create.argtypes = [C.c_float, C.c_int32, P(C.c_void_p)]
    L.cmx_grid3d_destroy.argtypes = [C.c_void_p]
    L.cmx_grid3d_destroy.restype = None
    L.cmx_grid3d_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float,
                                    C.c_float, C.c_int32]
    L.cmx_grid3d_info.argtypes = [C.c_void_p, P(C.c_float), P(C.c_int32), P(C.c_int64)]
    if hasattr(L, "cmx_intensity_grid3d_create"):      # (absent from the round-3 library)
        L.cmx_intensity_grid3d_create.argtypes = [C.c_float, C.c_int32, P(C.c_void_p)]
        L.cmx_intensity_grid3d_destroy.argtypes = [C.c_void_p]
        L.cmx_intensity_grid3d_destroy.restype = None
        L.cmx_grid3d_insert_with_intensities.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float,
            C.c_float, C.c_int32, C.c_float]
        L.cmx_intensity_grid3d_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64,
                                                    P(C.c_int64)]
        L.cmx_ceres3d_match_grids_intensity.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p]
    L.cmx_grid3d_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, P(C.c_int64)]
    L.cmx_grid2d_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_int32, C.c_float, C.c_float, C.c_int32]
    L.cmx_rt2d_match_grid.argtypes = [P(RtOptions), C.c_void_p, P(Pose2d), C.c_void_p, C.c_int32,
                                      P(C.c_double), P(Pose2d), P(MatchStats)]
    L.cmx_rt2d_match_grid_batch.argtypes = [P(RtOptions), P(C.c_void_p), C.c_int32, C.c_void_p,
                                            P(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p,
                                            P(MatchStats)]
    L.cmx_rt2d_match_grid_batch_resident.argtypes = [P(RtOptions), P(C.c_void_p), C.c_int32,
                                                     C.c_void_p, P(C.c_void_p), C.c_void_p,
                                                     C.c_void_p, P(MatchStats)]
    L.cmx_fast2d_create_from_grid.argtypes = [P(Fast2DOptions), C.c_void_p, P(C.c_void_p)]
    L.cmx_fast2d_create.argtypes = [P(Fast2DOptions), P(Grid2DLimits), C.c_void_p, C.c_int32,
                                    P(C.c_void_p)]
    L.cmx_fast2d_destroy.argtypes = [C.c_void_p]
    L.cmx_fast2d_destroy.restype = None
    L.cmx_fast2d_match.argtypes = [C.c_void_p, P(Pose2d), C.c_void_p, C.c_int32, C.c_float,
                                   P(C.c_int32), P(C.c_float), P(Pose2d), P(MatchStats)]
    L.cmx_fast2d_match_full_submap.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float,
                                               P(C.c_int32), P(C.c_float), P(Pose2d),
                                               P(MatchStats)]
    L.cmx_fast2d_match_full_submap_batch.argtypes = [P(C.c_void_p), C.c_int32, C.c_void_p,
                                                     C.c_int32, C.c_float, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, P(MatchStats)]
    L.cmx_fast2d_match_batch.argtypes = [P(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p, P(MatchStats)]
    L.cmx_ceres2d_match.argtypes = [P(Ceres2DOptions), P(Grid2DLimits), C.c_void_p, C.c_void_p,
                                    P(Pose2d), C.c_void_p, C.c_int32, C.c_int32, P(Pose2d),
                                    P(CeresSummary)]
    L.cmx_ceres2d_match_grid.argtypes = [P(Ceres2DOptions), C.c_void_p, C.c_void_p, P(Pose2d),
                                         C.c_void_p, C.c_int32, P(Pose2d), P(CeresSummary)]
    L.cmx_fast2d_refine_batch.argtypes = [P(Ceres2DOptions), P(C.c_void_p), C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p]
    L.cmx_ceres3d_match.argtypes = [P(Ceres3DOptions), C.c_void_p, P(Pose3d), C.c_void_p, C.c_int32,
                                    P(Pose3d), P(CeresSummary)]
    L.cmx_rt2d_score_candidates.argtypes = [P(RtOptions), P(Grid2DLimits), C.c_void_p, C.c_void_p,
                                            C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                            C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    L.cmx_rt3d_match_grid.argtypes = [P(RtOptions), C.c_void_p, P(Pose3d), C.c_void_p, C.c_int32,
                                      P(C.c_float), P(Pose3d), P(MatchStats)]
    L.cmx_ceres3d_match_grids.argtypes = [P(Ceres3DOptions), C.c_void_p, P(Pose3d), P(C.c_void_p),
                                          P(C.c_void_p), C.c_void_p, P(Pose3d), P(CeresSummary)]
    L.cmx_fast3d_refine_batch.argtypes = [P(Ceres3DOptions), P(C.c_void_p), C.c_int32, C.c_void_p,
                                          C.c_void_p, P(NodeData3D), C.c_void_p, C.c_void_p]
    L.cmx_voxel_filter.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p,
                                   P(C.c_int32)]
    L.cmx_voxel_filter_indices.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32,
                                           C.c_void_p, P(C.c_int32)]
    L.cmx_adaptive_voxel_filter.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                            C.c_float, C.c_int32, C.c_void_p, P(C.c_int32)]
    L.cmx_adaptive_voxel_filter_indices.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                                    C.c_float, C.c_int32, C.c_void_p, P(C.c_int32)]
    L.cmx_compute_histogram.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.cmx_comm_init.argtypes = [C.c_void_p, C.c_int32, P(C.c_void_p)]
    L.cmx_comm_destroy.argtypes = [C.c_void_p]
    L.cmx_comm_destroy.restype = None
    L.cmx_comm_num_devices.argtypes = [C.c_void_p]
    L.cmx_comm_num_devices.restype = C.c_int32
    L.cmx_comm_uses_rccl.argtypes = [C.c_void_p]
    L.cmx_comm_uses_rccl.restype = C.c_int32
    L.cmx_comm_device_of.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
    L.cmx_comm_device_of.restype = C.c_int32
    L.cmx_fast2d_match_sharded.argtypes = [C.c_void_p, P(C.c_void_p), C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int32),
                                           P(C.c_float), P(MatchStats)]
    L.cmx_fast3d_match_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, P(C.c_int32), P(C.c_float),
                                           P(MatchStats)]
    L.cmx_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, P(C.c_int64), P(C.c_int64)]
    L.cmx_shard_range.restype = None
    L.cmx_pack_best_key.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    L.cmx_pack_best_key.restype = C.c_int64
    L.cmx_unpack_best_key.argtypes = [C.c_int64, P(C.c_int32), P(C.c_float), P(C.c_int64)]
    L.cmx_unpack_best_key.restype = None
    L.cmx_cloud_upload.argtypes = [C.c_void_p, C.c_int32, C.c_int32, P(C.c_void_p)]
    L.cmx_cloud_destroy.argtypes = [C.c_void_p]
    L.cmx_cloud_destroy.restype = None
    L.cmx_fast2d_match_full_submap_batch_resident.argtypes = [
        P(C.c_void_p), C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
        P(MatchStats)]
    L.cmx_fast2d_level_dims.argtypes = [C.c_void_p, C.c_int32, P(C.c_int32), P(C.c_int32)]
    L.cmx_fast2d_level_cells.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.cmx_fast2d_debug_prepare.argtypes = [C.c_void_p, P(Pose2d), C.c_void_p, C.c_int32,
                                           C.c_int32, P(C.c_int32), P(C.c_double), C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int64, P(C.c_int64)]
    L.cmx_rt3d_match.argtypes = [P(RtOptions), C.c_float, C.c_void_p, C.c_int64, P(Pose3d),
                                 C.c_void_p, C.c_int32, C.c_int32, P(C.c_float), P(Pose3d),
                                 P(MatchStats)]
    L.cmx_fast3d_create.argtypes = [P(Fast3DOptions), C.c_float, C.c_int32, C.c_void_p, C.c_int64,
                                    C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                    C.c_int32, P(C.c_void_p)]
    L.cmx_fast3d_destroy.argtypes = [C.c_void_p]
    L.cmx_fast3d_destroy.restype = None
    L.cmx_fast3d_match.argtypes = [C.c_void_p, P(Pose3d), P(Pose3d), P(NodeData3D), C.c_float,
                                   P(C.c_int32), P(Result3D), P(MatchStats)]
    L.cmx_fast3d_match_full_submap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(NodeData3D),
                                               C.c_float, P(C.c_int32), P(Result3D),
                                               P(MatchStats)]
    L.cmx_fast3d_match_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, P(NodeData3D), C.c_void_p,
                                         C.c_void_p, P(MatchStats)]
    L.cmx_fast3d_level_info.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.cmx_fast3d_level_cells.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    if hasattr(L, "cmx_debug_set"):        # (absent from the round-3 library the A/B tools load)
        L.cmx_debug_set.argtypes = [C.c_char_p, C.c_int32]
        L.cmx_debug_reset.restype = None
    _lib = L
    return L


def debug_set(**switches):
    """cmx_debug_set for every keyword (tests and tools: which of two equivalent device paths runs,
    verification modes, tuning overrides).  Process-wide; `debug_reset()` puts everything back."""
    L = lib()
    for name, value in switches.items():
        check(L.cmx_debug_set(name.encode(), int(value)))


def debug_reset():
    lib().cmx_debug_reset()


def check(status):
    if status != OK:
        L = lib()
        raise CmxError(status, f"{L.cmx_status_string(status).decode()}: "
                               f"{L.cmx_last_error().decode()}")
