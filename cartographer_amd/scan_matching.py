"""Host-side mirror of the reference's scan-matcher classes over the C ABI.

Same names, argument meaning and failure behaviour as
``cartographer/mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d.h:48-83``
and ``.../fast_correlative_scan_matcher_2d.h:109-160`` (3D twins in
``.../3d/scan_matching``), so the parity tests read like the reference's own
tests.  All arithmetic happens in ``libcartographer_mi355x.so``.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (Ceres2DOptions, CeresSummary, Fast2DOptions, Grid2DLimits, MatchStats, Pose2d,
                   RtOptions, check)

K_MIN_CORRESPONDENCE_COST = float(np.float32(1) - (np.float32(1) - np.float32(0.1)))
K_MAX_CORRESPONDENCE_COST = float(np.float32(1) - np.float32(0.1))


@dataclass
class Rigid2d:
    """transform::Rigid2d: translation + rotation angle."""
    x: float = 0.0
    y: float = 0.0
    theta: float = 0.0

    def to_c(self):
        return Pose2d(self.x, self.y, self.theta)


class Grid2D:
    """Read-only view of a ProbabilityGrid: MapLimits + correspondence-cost cells."""

    def __init__(self, cells, resolution, max_x, max_y,
                 min_correspondence_cost=K_MIN_CORRESPONDENCE_COST,
                 max_correspondence_cost=K_MAX_CORRESPONDENCE_COST):
        self.cells = np.ascontiguousarray(cells, dtype=np.uint16)
        if self.cells.ndim != 2:
            raise ValueError("cells must be [num_y_cells, num_x_cells]")
        self.resolution = float(resolution)
        self.max_x = float(max_x)
        self.max_y = float(max_y)
        self.min_correspondence_cost = float(min_correspondence_cost)
        self.max_correspondence_cost = float(max_correspondence_cost)

    @property
    def num_x_cells(self):
        return self.cells.shape[1]

    @property
    def num_y_cells(self):
        return self.cells.shape[0]

    def limits_c(self):
        return Grid2DLimits(self.resolution, self.max_x, self.max_y, self.num_x_cells,
                            self.num_y_cells, self.min_correspondence_cost,
                            self.max_correspondence_cost)


class TSDF2D:
    """Read-only view of a TSDF2D (mapping/internal/2d/tsdf_2d.h): MapLimits, the tsd and
    weight planes (uint16, 0 = unknown) and the TSDValueConverter ranges."""

    def __init__(self, tsd_cells, weight_cells, resolution, max_x, max_y, truncation_distance,
                 max_weight):
        self.cells = np.ascontiguousarray(tsd_cells, dtype=np.uint16)
        self.weight_cells = np.ascontiguousarray(weight_cells, dtype=np.uint16)
        if self.cells.ndim != 2 or self.cells.shape != self.weight_cells.shape:
            raise ValueError("tsd / weight cells must both be [num_y_cells, num_x_cells]")
        self.resolution = float(resolution)
        self.max_x = float(max_x)
        self.max_y = float(max_y)
        self.truncation_distance = float(truncation_distance)
        self.max_weight = float(max_weight)

    def limits_c(self):
        # Grid2D(limits, -truncation_distance, truncation_distance) (tsdf_2d.cc:25-26)
        return Grid2DLimits(self.resolution, self.max_x, self.max_y, self.cells.shape[1],
                            self.cells.shape[0], -self.truncation_distance,
                            self.truncation_distance)


def _cloud(point_cloud):
    xyz = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(-1, 3)
    return xyz, xyz.shape[0]


class RealTimeCorrelativeScanMatcher2D:
    """RealTimeCorrelativeScanMatcher2D(options).Match(initial, cloud, grid) -> (score, pose)."""

    def __init__(self, linear_search_window, angular_search_window,
                 translation_delta_cost_weight, rotation_delta_cost_weight, device=0):
        self.options = RtOptions(linear_search_window, angular_search_window,
                                 translation_delta_cost_weight, rotation_delta_cost_weight)
        self.device = device
        self.last_stats = None

    def match(self, initial_pose_estimate, point_cloud, grid):
        xyz, n = _cloud(point_cloud)
        from .grid_2d import ProbabilityGridOnDevice
        if isinstance(grid, ProbabilityGridOnDevice):     # grid already in HBM
            init = initial_pose_estimate.to_c()
            score, pose, stats = C.c_double(), Pose2d(), MatchStats()
            check(_lib.lib().cmx_rt2d_match_grid(C.byref(self.options), grid._h, C.byref(init),
                                                 xyz.ctypes.data, n, C.byref(score),
                                                 C.byref(pose), C.byref(stats)))
            self.last_stats = stats.as_dict()
            return score.value, Rigid2d(pose.x, pose.y, pose.theta)
        limits = grid.limits_c()
        init = initial_pose_estimate.to_c()
        score = C.c_double()
        pose = Pose2d()
        stats = MatchStats()
        if isinstance(grid, TSDF2D):      # GridType::TSDF branch (.cc:159-167)
            check(_lib.lib().cmx_rt2d_match_tsdf(
                C.byref(self.options), C.byref(limits), grid.cells.ctypes.data,
                grid.weight_cells.ctypes.data, grid.truncation_distance, grid.max_weight,
                C.byref(init), xyz.ctypes.data, n, self.device, C.byref(score), C.byref(pose),
                C.byref(stats)))
        else:
            check(_lib.lib().cmx_rt2d_match(C.byref(self.options), C.byref(limits),
                                            grid.cells.ctypes.data, C.byref(init),
                                            xyz.ctypes.data, n, self.device, C.byref(score),
                                            C.byref(pose), C.byref(stats)))
        self.last_stats = stats.as_dict()
        return score.value, Rigid2d(pose.x, pose.y, pose.theta)


def rt2d_match_batch(options, grids, initial_pose_estimates, point_clouds):
    """cmx_rt2d_match_grid_batch: match i = (point_clouds[i], grids[i], initial_pose_estimates[i]);
    `grids` are ProbabilityGridOnDevice, `options` an RtOptions (or a
    RealTimeCorrelativeScanMatcher2D).  Returns (scores, poses, stats)."""
    if isinstance(options, RealTimeCorrelativeScanMatcher2D):
        options = options.options
    num = len(grids)
    clouds = [_cloud(c)[0] for c in point_clouds]
    handles = (C.c_void_p * num)(*[g._h for g in grids])
    cloud_ptrs = (C.c_void_p * num)(*[c.ctypes.data for c in clouds])
    counts = np.array([c.shape[0] for c in clouds], np.int32)
    initial = (Pose2d * num)(*[p.to_c() for p in initial_pose_estimates])
    scores = np.zeros(num, np.float64)
    poses = (Pose2d * num)()
    stats = MatchStats()
    check(_lib.lib().cmx_rt2d_match_grid_batch(
        C.byref(options), handles, num, C.cast(initial, C.c_void_p), cloud_ptrs,
        counts.ctypes.data, scores.ctypes.data, C.cast(poses, C.c_void_p), C.byref(stats)))
    return (scores, [Rigid2d(p.x, p.y, p.theta) for p in poses], stats.as_dict())


class Rt2DBatch:
    """The argument arrays of cmx_rt2d_match_grid_batch, built once: what a C++ caller holds
    anyway (one resident grid and one scan buffer per trajectory / robot).  `match(poses)` takes
    the initial pose estimates as an (n, 3) float64 array (x, y, theta) and returns
    (scores, poses as an (n, 3) array, stats)."""

    def __init__(self, options, grids, point_clouds, resident=False):
        """resident=True: the scans are uploaded once (cmx_cloud) and every match() goes through
        cmx_rt2d_match_grid_batch_resident; `point_clouds` may then also be PointCloudOnDevice."""
        if isinstance(options, RealTimeCorrelativeScanMatcher2D):
            options = options.options
        self.options = options
        self.num = len(grids)
        self._grids = list(grids)                                  # keep the handles alive
        self._handles = (C.c_void_p * self.num)(*[g._h for g in grids])
        self._scores = np.zeros(self.num, np.float64)
        self._poses = np.zeros((self.num, 3), np.float64)          # cmx_pose2d[num]
        self._stats = MatchStats()
        self.resident = resident
        if resident:
            self._clouds = [c if isinstance(c, PointCloudOnDevice) else PointCloudOnDevice(c)
                            for c in point_clouds]
            self._cloud_ptrs = (C.c_void_p * self.num)(*[c._h for c in self._clouds])
            self._fn = _lib.lib().cmx_rt2d_match_grid_batch_resident
        else:
            self._clouds = [_cloud(c)[0] for c in point_clouds]
            self._cloud_ptrs = (C.c_void_p * self.num)(*[c.ctypes.data for c in self._clouds])
            self._counts = np.array([c.shape[0] for c in self._clouds], np.int32)
            self._fn = _lib.lib().cmx_rt2d_match_grid_batch

    def match(self, initial_pose_estimates):
        init = np.ascontiguousarray(initial_pose_estimates, np.float64).reshape(self.num, 3)
        if self.resident:
            check(self._fn(C.byref(self.options), self._handles, self.num, init.ctypes.data,
                           self._cloud_ptrs, self._scores.ctypes.data, self._poses.ctypes.data,
                           C.byref(self._stats)))
        else:
            check(self._fn(C.byref(self.options), self._handles, self.num, init.ctypes.data,
                           self._cloud_ptrs, self._counts.ctypes.data, self._scores.ctypes.data,
                           self._poses.ctypes.data, C.byref(self._stats)))
        return self._scores, self._poses, self._stats.as_dict()


class PointCloudOnDevice:
    """A point cloud uploaded once (cmx_cloud) for repeated resident matches."""

    def __init__(self, point_cloud, device=0):
        xyz, n = _cloud(point_cloud)
        self.num_points = n
        self._h = C.c_void_p()
        check(_lib.lib().cmx_cloud_upload(xyz.ctypes.data, n, device, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_cloud_destroy(self._h)
            self._h = None


class FastCorrelativeScanMatcher2D:
    """FastCorrelativeScanMatcher2D(grid, options): Match / MatchFullSubmap.

    ``match*`` return ``(found, score, pose)``; ``found`` False mirrors the
    reference returning ``false`` (score/pose are then None).
    """

    def __init__(self, grid, branch_and_bound_depth, linear_search_window=7.0,
                 angular_search_window=float(np.deg2rad(30.0)), device=0):
        self.grid = grid
        self.options = Fast2DOptions(linear_search_window, angular_search_window,
                                     branch_and_bound_depth)
        self.device = device
        self.last_stats = None
        self._h = C.c_void_p()
        limits = grid.limits_c()
        check(_lib.lib().cmx_fast2d_create(C.byref(self.options), C.byref(limits),
                                           grid.cells.ctypes.data, device, C.byref(self._h)))

    @classmethod
    def from_device_grid(cls, device_grid, branch_and_bound_depth, linear_search_window=7.0,
                         angular_search_window=float(np.deg2rad(30.0))):
        """Matcher of a grid that lives in HBM (cartographer_amd.grid_2d.ProbabilityGridOnDevice)."""
        self = cls.__new__(cls)
        self.grid = device_grid
        self.options = Fast2DOptions(linear_search_window, angular_search_window,
                                     branch_and_bound_depth)
        self.device = device_grid.device
        self.last_stats = None
        self._h = C.c_void_p()
        check(_lib.lib().cmx_fast2d_create_from_grid(C.byref(self.options), device_grid._h,
                                                     C.byref(self._h)))
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_fast2d_destroy(self._h)
            self._h = None

    def _result(self, found, score, pose, stats):
        self.last_stats = stats.as_dict()
        if not found.value:
            return False, None, None
        return True, float(score.value), Rigid2d(pose.x, pose.y, pose.theta)

    def match(self, initial_pose_estimate, point_cloud, min_score):
        xyz, n = _cloud(point_cloud)
        init = initial_pose_estimate.to_c()
        found, score, pose, stats = C.c_int32(), C.c_float(), Pose2d(), MatchStats()
        check(_lib.lib().cmx_fast2d_match(self._h, C.byref(init), xyz.ctypes.data, n, min_score,
                                          C.byref(found), C.byref(score), C.byref(pose),
                                          C.byref(stats)))
        return self._result(found, score, pose, stats)

    def match_full_submap(self, point_cloud, min_score):
        xyz, n = _cloud(point_cloud)
        found, score, pose, stats = C.c_int32(), C.c_float(), Pose2d(), MatchStats()
        check(_lib.lib().cmx_fast2d_match_full_submap(self._h, xyz.ctypes.data, n, min_score,
                                                      C.byref(found), C.byref(score),
                                                      C.byref(pose), C.byref(stats)))
        return self._result(found, score, pose, stats)

    # -- introspection for the parity tests ---------------------------------
    def level(self, i):
        wx, wy = C.c_int32(), C.c_int32()
        check(_lib.lib().cmx_fast2d_level_dims(self._h, i, C.byref(wx), C.byref(wy)))
        out = np.empty((wy.value, wx.value), np.uint8)
        check(_lib.lib().cmx_fast2d_level_cells(self._h, i, out.ctypes.data))
        return out

    def debug_prepare(self, initial_pose_estimate, point_cloud, full_submap):
        xyz, n = _cloud(point_cloud)
        init = (initial_pose_estimate or Rigid2d()).to_c()
        ns, step, ncoarse = C.c_int32(), C.c_double(), C.c_int64()
        L = _lib.lib()
        check(L.cmx_fast2d_debug_prepare(self._h, C.byref(init), xyz.ctypes.data, n,
                                         int(full_submap), C.byref(ns), C.byref(step), None, 0,
                                         None, 0, None, 0, C.byref(ncoarse)))
        scans = np.empty((ns.value, n, 2), np.int32)
        bounds = np.empty((ns.value, 4), np.int32)
        sums = np.empty(ncoarse.value, np.int32)
        check(L.cmx_fast2d_debug_prepare(self._h, C.byref(init), xyz.ctypes.data, n,
                                         int(full_submap), C.byref(ns), C.byref(step),
                                         scans.ctypes.data, scans.size, bounds.ctypes.data,
                                         bounds.size, sums.ctypes.data, sums.size,
                                         C.byref(ncoarse)))
        return dict(num_scans=ns.value, step=step.value, scans=scans, bounds=bounds, sums=sums)


def match_full_submap_batch(matchers, point_cloud, min_score):
    """One scan against many submaps: the ConstraintBuilder2D fan-out
    (constraints/constraint_builder_2d.cc:97-137) as a single device batch.

    ``point_cloud`` is an array or a PointCloudOnDevice.  Returns
    (found[int32], scores[float32], poses[n,3], stats dict).
    """
    num = len(matchers)
    handles = (C.c_void_p * num)(*[m._h for m in matchers])
    found = np.zeros(num, np.int32)
    scores = np.zeros(num, np.float32)
    poses = np.zeros((num, 3), np.float64)
    stats = MatchStats()
    L = _lib.lib()
    if isinstance(point_cloud, PointCloudOnDevice):
        check(L.cmx_fast2d_match_full_submap_batch_resident(
            handles, num, point_cloud._h, min_score, found.ctypes.data, scores.ctypes.data,
            poses.ctypes.data, C.byref(stats)))
    else:
        xyz, n = _cloud(point_cloud)
        check(L.cmx_fast2d_match_full_submap_batch(handles, num, xyz.ctypes.data, n, min_score,
                                                   found.ctypes.data, scores.ctypes.data,
                                                   poses.ctypes.data, C.byref(stats)))
    return found, scores, poses, stats.as_dict()


def match_batch(matchers, initial_pose_estimates, match_full_submap, min_scores, point_cloud):
    """cmx_fast2d_match_batch: one node's scan against many submaps, entry i either a windowed
    Match around initial_pose_estimates[i] (match_full_submap[i] == 0, MaybeAddConstraint) or a
    MatchFullSubmap (!= 0, MaybeAddGlobalConstraint), each against its own threshold
    (constraints/constraint_builder_2d.cc:77-137, :194-236).  Returns
    (found[int32], scores[float32], poses[list of Rigid2d], stats dict)."""
    num = len(matchers)
    handles = (C.c_void_p * num)(*[m._h for m in matchers])
    initial = (Pose2d * num)(*[p.to_c() for p in initial_pose_estimates])
    full = np.ascontiguousarray(match_full_submap, np.int32)
    thresholds = np.ascontiguousarray(min_scores, np.float32)
    xyz, n = _cloud(point_cloud)
    found = np.zeros(num, np.int32)
    scores = np.zeros(num, np.float32)
    poses = (Pose2d * num)()
    stats = MatchStats()
    check(_lib.lib().cmx_fast2d_match_batch(
        handles, num, C.cast(initial, C.c_void_p), full.ctypes.data, thresholds.ctypes.data,
        xyz.ctypes.data, n, found.ctypes.data, scores.ctypes.data, C.cast(poses, C.c_void_p),
        C.byref(stats)))
    return found, scores, [Rigid2d(p.x, p.y, p.theta) for p in poses], stats.as_dict()


class CeresScanMatcher2D:
    """CeresScanMatcher2D(options).Match(target_translation, initial_pose_estimate, point_cloud,
    grid) -> (pose_estimate, summary)  (ceres_scan_matcher_2d.h:44-56).  `grid` is a Grid2D or a
    grid resident in HBM (cartographer_amd.grid_2d.ProbabilityGridOnDevice)."""

    def __init__(self, occupied_space_weight, translation_weight, rotation_weight,
                 use_nonmonotonic_steps=False, max_num_iterations=20, device=0):
        self.options = Ceres2DOptions(occupied_space_weight, translation_weight, rotation_weight,
                                      1 if use_nonmonotonic_steps else 0, max_num_iterations)
        self.device = device

    def match(self, target_translation, initial_pose_estimate, point_cloud, grid):
        from .grid_2d import ProbabilityGridOnDevice
        xyz, n = _cloud(point_cloud)
        target = np.ascontiguousarray(target_translation, np.float64)
        init = initial_pose_estimate.to_c()
        pose, summary = Pose2d(), CeresSummary()
        if isinstance(grid, ProbabilityGridOnDevice):
            check(_lib.lib().cmx_ceres2d_match_grid(C.byref(self.options), grid._h,
                                                    target.ctypes.data, C.byref(init),
                                                    xyz.ctypes.data, n, C.byref(pose),
                                                    C.byref(summary)))
        else:
            limits = grid.limits_c()
            check(_lib.lib().cmx_ceres2d_match(C.byref(self.options), C.byref(limits),
                                               grid.cells.ctypes.data, target.ctypes.data,
                                               C.byref(init), xyz.ctypes.data, n, self.device,
                                               C.byref(pose), C.byref(summary)))
        return Rigid2d(pose.x, pose.y, pose.theta), summary.as_dict()

    def refine_batch(self, matchers, found, pose_estimates, point_cloud):
        """ConstraintBuilder2D::ComputeConstraint's refinement of a batch of search results
        (constraint_builder_2d.cc:245-249), each against the grid its matcher keeps in HBM."""
        num = len(matchers)
        xyz, n = _cloud(point_cloud)
        handles = (C.c_void_p * num)(*[m._h for m in matchers])
        found = np.ascontiguousarray(found, np.int32)
        poses_in = (Pose2d * num)(*[p.to_c() for p in pose_estimates])
        poses_out = (Pose2d * num)()
        summaries = (CeresSummary * num)()
        check(_lib.lib().cmx_fast2d_refine_batch(
            C.byref(self.options), handles, num, found.ctypes.data, C.cast(poses_in, C.c_void_p),
            xyz.ctypes.data, n, C.cast(poses_out, C.c_void_p), C.cast(summaries, C.c_void_p)))
        return ([Rigid2d(p.x, p.y, p.theta) for p in poses_out],
                [s_.as_dict() for s_ in summaries])
