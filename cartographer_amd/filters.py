"""Host-side mirror of the reference's point preparation over the C ABI:
``sensor::VoxelFilter`` / ``sensor::AdaptiveVoxelFilter``
(cartographer/sensor/internal/voxel_filter.h:30-45) and
``RotationalScanMatcher::ComputeHistogram``
(cartographer/mapping/internal/3d/scan_matching/rotational_scan_matcher.h:40-42)."""
import ctypes as C

import numpy as np

from . import _lib


def _cloud(point_cloud):
    xyz = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(-1, 3)
    return xyz, xyz.shape[0]


def voxel_filter(point_cloud, resolution, device=0):
    xyz, n = _cloud(point_cloud)
    out = np.empty((max(n, 1), 3), np.float32)
    kept = C.c_int32()
    _lib.check(_lib.lib().cmx_voxel_filter(xyz.ctypes.data, n, resolution, device,
                                           out.ctypes.data, C.byref(kept)))
    return out[:kept.value].copy()


def voxel_filter_indices(point_cloud, resolution, device=0):
    """Which points ``voxel_filter`` keeps (ascending indices): the overloads of
    ``sensor::VoxelFilter`` over timed points / range measurements
    (cartographer/sensor/internal/voxel_filter.cc:154-191) select payload with them."""
    xyz, n = _cloud(point_cloud)
    out = np.empty(max(n, 1), np.int32)
    kept = C.c_int32()
    _lib.check(_lib.lib().cmx_voxel_filter_indices(xyz.ctypes.data, n, resolution, device,
                                                   out.ctypes.data, C.byref(kept)))
    return out[:kept.value].copy()


def adaptive_voxel_filter(point_cloud, max_length, min_num_points, max_range, device=0):
    xyz, n = _cloud(point_cloud)
    out = np.empty((max(n, 1), 3), np.float32)
    kept = C.c_int32()
    _lib.check(_lib.lib().cmx_adaptive_voxel_filter(xyz.ctypes.data, n, max_length,
                                                    min_num_points, max_range, device,
                                                    out.ctypes.data, C.byref(kept)))
    return out[:kept.value].copy()


def adaptive_voxel_filter_indices(point_cloud, max_length, min_num_points, max_range, device=0):
    """Which points ``adaptive_voxel_filter`` keeps (ascending indices into the input): a
    ``sensor::PointCloud`` keeps the intensities of the kept points
    (cartographer/sensor/internal/voxel_filter.cc:138-161,193-198)."""
    xyz, n = _cloud(point_cloud)
    out = np.empty(max(n, 1), np.int32)
    kept = C.c_int32()
    _lib.check(_lib.lib().cmx_adaptive_voxel_filter_indices(xyz.ctypes.data, n, max_length,
                                                            min_num_points, max_range, device,
                                                            out.ctypes.data, C.byref(kept)))
    return out[:kept.value].copy()


def compute_histogram(point_cloud, histogram_size, device=0):
    xyz, n = _cloud(point_cloud)
    out = np.zeros(histogram_size, np.float32)
    _lib.check(_lib.lib().cmx_compute_histogram(xyz.ctypes.data, n, histogram_size, device,
                                                out.ctypes.data))
    return out
