"""Device-resident ProbabilityGrid: range-data insertion and real-time
matching without moving the grid across PCIe.

Mirrors what LocalTrajectoryBuilder2D does with the active submap
(``mapping/internal/2d/local_trajectory_builder_2d.cc:78-80, 288-289``):
``ProbabilityGridRangeDataInserter2D::Insert`` (``insert``) and
``RealTimeCorrelativeScanMatcher2D::Match`` (``RealTimeCorrelativeScanMatcher2D.match`` accepts
this class in place of a host ``Grid2D``).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Grid2DLimits, check


class ProbabilityGridOnDevice:
    def __init__(self, resolution, max_xy, num_x_cells, num_y_cells, cells=None, device=0):
        limits = Grid2DLimits(resolution, max_xy[0], max_xy[1], num_x_cells, num_y_cells, 0.0, 0.0)
        self.device = device
        self._h = C.c_void_p()
        ptr = None
        if cells is not None:
            cells = np.ascontiguousarray(cells, np.uint16)
            assert cells.shape == (num_y_cells, num_x_cells)
            ptr = cells.ctypes.data
        check(_lib.lib().cmx_grid2d_create(C.byref(limits), ptr, device, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_grid2d_destroy(self._h)
            self._h = None

    @property
    def limits(self):
        lim = Grid2DLimits()
        check(_lib.lib().cmx_grid2d_get_limits(self._h, C.byref(lim)))
        return dict(resolution=lim.resolution, max_x=lim.max_x, max_y=lim.max_y,
                    num_x_cells=lim.num_x_cells, num_y_cells=lim.num_y_cells)

    @property
    def cells(self):
        lim = self.limits
        out = np.empty((lim["num_y_cells"], lim["num_x_cells"]), np.uint16)
        check(_lib.lib().cmx_grid2d_download(self._h, out.ctypes.data))
        return out

    def insert(self, origin_xy, returns_xyz, misses_xyz=None, hit_probability=0.7,
               miss_probability=0.4, insert_free_space=True):
        """ProbabilityGridRangeDataInserter2D::Insert + FinishUpdate; points in the map frame."""
        origin = np.ascontiguousarray(origin_xy, np.float32)[:2].copy()
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        mis = (np.ascontiguousarray(misses_xyz, np.float32).reshape(-1, 3)
               if misses_xyz is not None else np.zeros((0, 3), np.float32))
        check(_lib.lib().cmx_grid2d_insert(
            self._h, origin.ctypes.data, ret.ctypes.data if ret.shape[0] else None, ret.shape[0],
            mis.ctypes.data if mis.shape[0] else None, mis.shape[0], hit_probability,
            miss_probability, int(insert_free_space)))

    def crop(self):
        """grid = grid->ComputeCroppedGrid() (probability_grid.cc:90-106): shrinks the grid to the
        bounding box of its known cells, as Submap2D::Finish does before the loop-closure matcher
        of the submap is built."""
        check(_lib.lib().cmx_grid2d_crop(self._h))

    def fast_matcher(self, branch_and_bound_depth, linear_search_window=7.0,
                     angular_search_window=float(np.deg2rad(30.0))):
        """FastCorrelativeScanMatcher2D of this (finished) grid."""
        from .scan_matching import FastCorrelativeScanMatcher2D
        return FastCorrelativeScanMatcher2D.from_device_grid(
            self, branch_and_bound_depth, linear_search_window, angular_search_window)
