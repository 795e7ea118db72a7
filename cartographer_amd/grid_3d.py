"""Device-resident HybridGrid: range-data insertion without moving the
grid across PCIe.

Mirrors what LocalTrajectoryBuilder3D does with an active submap's grids:
``RangeDataInserter3D::Insert`` (``mapping/3d/range_data_inserter_3d.cc:93-114``, ``insert``);
``voxels()`` is the flattened list ``HybridGrid::Iterator`` yields, which is what
``scan_matching_3d`` takes for matching.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import INTENSITY_VOXEL_DTYPE, VOXEL_DTYPE, check


class HybridGridOnDevice:
    def __init__(self, resolution, device=0):
        self.device = device
        self._h = C.c_void_p()
        check(_lib.lib().cmx_grid3d_create(resolution, device, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_grid3d_destroy(self._h)
            self._h = None

    def _info(self):
        res, size, count = C.c_float(), C.c_int32(), C.c_int64()
        check(_lib.lib().cmx_grid3d_info(self._h, C.byref(res), C.byref(size), C.byref(count)))
        return float(res.value), int(size.value), int(count.value)

    @property
    def resolution(self):
        return self._info()[0]

    @property
    def grid_size(self):
        """DynamicGrid::grid_size() after everything written so far."""
        return self._info()[1]

    def insert(self, origin_xyz, returns_xyz, hit_probability=0.7, miss_probability=0.4,
               num_free_space_voxels=5):
        """RangeDataInserter3D::Insert (no intensities); points in the map frame."""
        origin = np.ascontiguousarray(origin_xyz, np.float32).reshape(3).copy()
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        check(_lib.lib().cmx_grid3d_insert(
            self._h, origin.ctypes.data, ret.ctypes.data if ret.shape[0] else None, ret.shape[0],
            hit_probability, miss_probability, num_free_space_voxels))

    def insert_with_intensities(self, intensity_grid, origin_xyz, returns_xyz, intensities,
                                hit_probability=0.7, miss_probability=0.4, num_free_space_voxels=5,
                                intensity_threshold=40.0):
        """RangeDataInserter3D::Insert with an intensity_hybrid_grid
        (range_data_inserter_3d.cc:93-114): this grid as ``insert``, then
        InsertIntensitiesIntoGrid (:54-70) into ``intensity_grid``
        (``IntensityHybridGridOnDevice``).  ``intensities``: one per return, or None."""
        origin = np.ascontiguousarray(origin_xyz, np.float32).reshape(3).copy()
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        ints = None if intensities is None else np.ascontiguousarray(intensities, np.float32)
        assert ints is None or ints.shape[0] == ret.shape[0]
        check(_lib.lib().cmx_grid3d_insert_with_intensities(
            self._h, intensity_grid._h, origin.ctypes.data,
            ret.ctypes.data if ret.shape[0] else None, None if ints is None else ints.ctypes.data,
            ret.shape[0], hit_probability, miss_probability, num_free_space_voxels,
            intensity_threshold))

    def voxels(self):
        """Known voxels as a VOXEL_DTYPE array sorted (z, y, x)."""
        count = self._info()[2]
        out = np.zeros(count, VOXEL_DTYPE)
        got = C.c_int64()
        check(_lib.lib().cmx_grid3d_download(self._h, out.ctypes.data if count else None, count,
                                             C.byref(got)))
        assert got.value == count
        return out


class IntensityHybridGridOnDevice:
    """Device-resident IntensityHybridGrid (mapping/3d/hybrid_grid.h:543-571): per voxel the
    AverageIntensityData {sum, count}; filled by ``HybridGridOnDevice.insert_with_intensities``,
    read in place by ``scan_matching_3d.CeresScanMatcher3D.match_grids``."""

    def __init__(self, resolution, device=0):
        self.device = device
        self._h = C.c_void_p()
        check(_lib.lib().cmx_intensity_grid3d_create(resolution, device, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_intensity_grid3d_destroy(self._h)
            self._h = None

    def voxels(self):
        """Voxels with count > 0 as an INTENSITY_VOXEL_DTYPE array sorted (z, y, x)."""
        got = C.c_int64()
        check(_lib.lib().cmx_intensity_grid3d_download(self._h, None, 0, C.byref(got)))
        out = np.zeros(got.value, INTENSITY_VOXEL_DTYPE)
        check(_lib.lib().cmx_intensity_grid3d_download(
            self._h, out.ctypes.data if got.value else None, got.value, C.byref(got)))
        return out
