"""Host-side mirror of the reference's 3D scan-matcher classes over the C ABI
(``cartographer/mapping/internal/3d/scan_matching/real_time_correlative_scan_matcher_3d.h:36-62``
and ``.../fast_correlative_scan_matcher_3d.h:54-137``).

A HybridGrid is handed over as the flattened voxel list its Iterator yields
(numpy structured array ``_lib.VOXEL_DTYPE``); poses are ``Rigid3d`` =
translation + quaternion (w, x, y, z).
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import (Ceres3DOptions, Ceres3DPair, CeresSummary, Fast3DOptions, MatchStats, NodeData3D,
                   Pose3d, Result3D, RtOptions, VOXEL_DTYPE, check)


@dataclass
class Rigid3d:
    translation: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (1.0, 0.0, 0.0, 0.0)   # w, x, y, z

    def to_c(self):
        p = Pose3d()
        p.t[:] = [float(v) for v in self.translation]
        p.q[:] = [float(v) for v in self.rotation]
        return p

    @staticmethod
    def from_c(p):
        return Rigid3d(tuple(p.t), tuple(p.q))

    def as_array(self):
        return np.array(list(self.translation) + list(self.rotation), np.float64)


def _voxels(v):
    v = np.ascontiguousarray(v, dtype=VOXEL_DTYPE)
    return v, v.shape[0]


def _cloud(point_cloud):
    xyz = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(-1, 3)
    return xyz, xyz.shape[0]


class RealTimeCorrelativeScanMatcher3D:
    """Match(initial_pose, cloud, hybrid_grid) -> (score, pose)."""

    def __init__(self, linear_search_window, angular_search_window,
                 translation_delta_cost_weight, rotation_delta_cost_weight, device=0):
        self.options = RtOptions(linear_search_window, angular_search_window,
                                 translation_delta_cost_weight, rotation_delta_cost_weight)
        self.device = device
        self.last_stats = None

    def match_grid(self, initial_pose_estimate, point_cloud, grid):
        """``match`` against a HybridGrid that already lives in HBM
        (``grid_3d.HybridGridOnDevice``): only the scan crosses PCIe."""
        xyz, n = _cloud(point_cloud)
        init = initial_pose_estimate.to_c()
        score, pose, stats = C.c_float(), Pose3d(), MatchStats()
        check(_lib.lib().cmx_rt3d_match_grid(C.byref(self.options), grid._h, C.byref(init),
                                             xyz.ctypes.data, n, C.byref(score), C.byref(pose),
                                             C.byref(stats)))
        self.last_stats = stats.as_dict()
        return float(score.value), Rigid3d.from_c(pose)

    def match(self, initial_pose_estimate, point_cloud, grid_resolution, grid_voxels):
        vox, nv = _voxels(grid_voxels)
        xyz, n = _cloud(point_cloud)
        init = initial_pose_estimate.to_c()
        score = C.c_float()
        pose = Pose3d()
        stats = MatchStats()
        check(_lib.lib().cmx_rt3d_match(C.byref(self.options), grid_resolution, vox.ctypes.data,
                                        nv, C.byref(init), xyz.ctypes.data, n, self.device,
                                        C.byref(score), C.byref(pose), C.byref(stats)))
        self.last_stats = stats.as_dict()
        return float(score.value), Rigid3d.from_c(pose)


class CeresScanMatcher3D:
    """CeresScanMatcher3D (SM3/ceres_scan_matcher_3d.h:48-66).

    ``match(target_translation, initial_pose_estimate, point_clouds_and_hybrid_grids)`` with
    ``point_clouds_and_hybrid_grids = [(point_cloud, grid_resolution, grid_voxels), ...]`` (one
    entry per occupied_space_weight) returns ``(pose_estimate, summary dict)``.  An entry may
    continue with ``intensities, intensity_voxels, (weight, huber_scale, intensity_threshold)``:
    the pair's intensity_hybrid_grid with its IntensityCostFunctionOptions."""

    def __init__(self, occupied_space_weights, translation_weight, rotation_weight,
                 only_optimize_yaw=False, use_nonmonotonic_steps=False, max_num_iterations=12,
                 device=0):
        o = Ceres3DOptions()
        assert 1 <= len(occupied_space_weights) <= 3
        for k, w in enumerate(occupied_space_weights):
            o.occupied_space_weight[k] = float(w)
        o.num_pairs = len(occupied_space_weights)
        o.translation_weight = float(translation_weight)
        o.rotation_weight = float(rotation_weight)
        o.only_optimize_yaw = 1 if only_optimize_yaw else 0
        o.use_nonmonotonic_steps = 1 if use_nonmonotonic_steps else 0
        o.max_num_iterations = int(max_num_iterations)
        self.options = o
        self.device = device

    def match(self, target_translation, initial_pose_estimate, point_clouds_and_hybrid_grids):
        num = self.options.num_pairs
        assert len(point_clouds_and_hybrid_grids) == num
        keep = []
        pairs = (Ceres3DPair * num)()
        for k, entry in enumerate(point_clouds_and_hybrid_grids):
            cloud, resolution, voxels = entry[:3]
            xyz, n = _cloud(cloud)
            vox, nv = _voxels(voxels)
            keep += [xyz, vox]
            pairs[k].point_cloud_xyz = xyz.ctypes.data
            pairs[k].num_points = n
            pairs[k].resolution = float(resolution)
            pairs[k].voxels = vox.ctypes.data if nv else None
            pairs[k].num_voxels = nv
            if len(entry) > 3 and entry[3] is not None:
                # (intensities, intensity voxels, (weight, huber_scale, intensity_threshold)):
                # the pair's intensity_hybrid_grid and its IntensityCostFunctionOptions
                from ._lib import INTENSITY_VOXEL_DTYPE
                ints = np.ascontiguousarray(entry[3], np.float32)
                ivox = np.ascontiguousarray(entry[4], INTENSITY_VOXEL_DTYPE)
                keep += [ints, ivox]
                assert ints.shape[0] == n
                pairs[k].intensities = ints.ctypes.data
                pairs[k].intensity_voxels = ivox.ctypes.data if ivox.shape[0] else None
                pairs[k].num_intensity_voxels = ivox.shape[0]
                pairs[k].intensity_weight = float(entry[5][0])
                pairs[k].intensity_huber_scale = float(entry[5][1])
                pairs[k].intensity_threshold = float(entry[5][2])
        target = np.ascontiguousarray(target_translation, np.float64)
        init = initial_pose_estimate.to_c()
        pose = Pose3d()
        summary = CeresSummary()
        check(_lib.lib().cmx_ceres3d_match(C.byref(self.options), target.ctypes.data, C.byref(init),
                                           C.cast(pairs, C.c_void_p), self.device, C.byref(pose),
                                           C.byref(summary)))
        del keep
        return Rigid3d.from_c(pose), summary.as_dict()

    def match_grids(self, target_translation, initial_pose_estimate, point_clouds_and_grids):
        """``match`` against HybridGrids that already live in HBM
        (``grid_3d.HybridGridOnDevice``): ``[(point_cloud, grid), ...]`` -- what
        LocalTrajectoryBuilder3D::ScanMatch does with the active submap
        (local_trajectory_builder_3d.cc:96-123).  Only the clouds cross PCIe.  An entry may
        continue with ``intensities, intensity_grid, (weight, huber_scale, intensity_threshold)``:
        the pair's resident ``grid_3d.IntensityHybridGridOnDevice``."""
        num = self.options.num_pairs
        assert len(point_clouds_and_grids) == num
        clouds = [_cloud(e[0])[0] for e in point_clouds_and_grids]
        handles = (C.c_void_p * num)(*[e[1]._h for e in point_clouds_and_grids])
        if any(len(e) > 2 and e[2] is not None for e in point_clouds_and_grids):
            from ._lib import Ceres3DIntensityTerm
            terms = (Ceres3DIntensityTerm * num)()
            keep = []
            for k, e in enumerate(point_clouds_and_grids):
                if len(e) <= 2 or e[2] is None:
                    continue
                ints = np.ascontiguousarray(e[2], np.float32)
                assert ints.shape[0] == clouds[k].shape[0]
                keep.append(ints)
                terms[k].grid = e[3]._h
                terms[k].intensities = ints.ctypes.data
                terms[k].weight, terms[k].huber_scale, terms[k].intensity_threshold = \
                    float(e[4][0]), float(e[4][1]), float(e[4][2])
            pointers = (C.c_void_p * num)(*[c.ctypes.data for c in clouds])
            counts = np.ascontiguousarray([c.shape[0] for c in clouds], np.int32)
            target = np.ascontiguousarray(target_translation, np.float64)
            init = initial_pose_estimate.to_c()
            pose = Pose3d()
            summary = CeresSummary()
            check(_lib.lib().cmx_ceres3d_match_grids_intensity(
                C.byref(self.options), target.ctypes.data, C.byref(init), handles, pointers,
                counts.ctypes.data, C.cast(terms, C.c_void_p), C.byref(pose), C.byref(summary)))
            del keep
            return Rigid3d.from_c(pose), summary.as_dict()
        pointers = (C.c_void_p * num)(*[c.ctypes.data for c in clouds])
        counts = np.ascontiguousarray([c.shape[0] for c in clouds], np.int32)
        target = np.ascontiguousarray(target_translation, np.float64)
        init = initial_pose_estimate.to_c()
        pose = Pose3d()
        summary = CeresSummary()
        check(_lib.lib().cmx_ceres3d_match_grids(C.byref(self.options), target.ctypes.data,
                                                 C.byref(init), handles, pointers,
                                                 counts.ctypes.data, C.byref(pose),
                                                 C.byref(summary)))
        return Rigid3d.from_c(pose), summary.as_dict()

    def refine_batch(self, matchers, found, pose_estimates, constant_data):
        """ConstraintBuilder3D::ComputeConstraint's refinement of a node's search results
        (constraint_builder_3d.cc:263-276): entry i against the high- and low-resolution grids
        ``matchers[i]`` keeps in HBM, one launch.  Returns (poses, summaries); entries with
        ``found[i] == 0`` are passed through."""
        num = len(matchers)
        assert self.options.num_pairs == 2
        handles = (C.c_void_p * num)(*[m._h for m in matchers])
        found = np.ascontiguousarray([1 if f else 0 for f in found], np.int32)
        poses_in = (Pose3d * num)(*[p.to_c() for p in pose_estimates])
        poses_out = (Pose3d * num)()
        summaries = (CeresSummary * num)()
        data = constant_data.to_c()
        check(_lib.lib().cmx_fast3d_refine_batch(
            C.byref(self.options), handles, num, found.ctypes.data, C.cast(poses_in, C.c_void_p),
            C.byref(data), C.cast(poses_out, C.c_void_p), C.cast(summaries, C.c_void_p)))
        return [Rigid3d.from_c(p) for p in poses_out], [s.as_dict() for s in summaries]


@dataclass
class TrajectoryNodeData:
    """TrajectoryNode::Data fields the 3D matcher reads (mapping/trajectory_node.h:45-63)."""
    high_resolution_point_cloud: np.ndarray
    low_resolution_point_cloud: np.ndarray
    rotational_scan_matcher_histogram: np.ndarray
    gravity_alignment: tuple = (1.0, 0.0, 0.0, 0.0)
    _keep: list = field(default_factory=list, repr=False)

    def to_c(self):
        hi, nhi = _cloud(self.high_resolution_point_cloud)
        lo, nlo = _cloud(self.low_resolution_point_cloud)
        hist = np.ascontiguousarray(self.rotational_scan_matcher_histogram, np.float32)
        self._keep[:] = [hi, lo, hist]
        d = NodeData3D()
        d.gravity_alignment[:] = [float(v) for v in self.gravity_alignment]
        d.high_resolution_point_cloud = hi.ctypes.data
        d.num_high_resolution_points = nhi
        d.low_resolution_point_cloud = lo.ctypes.data
        d.num_low_resolution_points = nlo
        d.rotational_scan_matcher_histogram = hist.ctypes.data if hist.size else None
        d.histogram_size = hist.shape[0]
        return d


class FastCorrelativeScanMatcher3D:
    """FastCorrelativeScanMatcher3D(hybrid_grid, low_resolution_grid, histogram, options).

    ``match*`` return a dict (score, pose_estimate, rotational_score,
    low_resolution_score) or None — the reference's ``unique_ptr<Result>``.
    """

    def __init__(self, resolution, voxels, grid_size, low_resolution, low_resolution_voxels,
                 rotational_scan_matcher_histogram, branch_and_bound_depth=8,
                 full_resolution_depth=3, min_rotational_score=0.77,
                 min_low_resolution_score=0.55, linear_xy_search_window=5.0,
                 linear_z_search_window=1.0, angular_search_window=float(np.deg2rad(15.0)),
                 device=0):
        self.options = Fast3DOptions(branch_and_bound_depth, full_resolution_depth,
                                     min_rotational_score, min_low_resolution_score,
                                     linear_xy_search_window, linear_z_search_window,
                                     angular_search_window)
        vox, nv = _voxels(voxels)
        low, nl = _voxels(low_resolution_voxels)
        hist = np.ascontiguousarray(rotational_scan_matcher_histogram, np.float32)
        self.depth = branch_and_bound_depth
        self.last_stats = None
        self._h = C.c_void_p()
        check(_lib.lib().cmx_fast3d_create(C.byref(self.options), resolution, grid_size,
                                           vox.ctypes.data, nv, low_resolution, low.ctypes.data,
                                           nl, hist.ctypes.data if hist.size else None,
                                           hist.shape[0], device, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().cmx_fast3d_destroy(self._h)
            self._h = None

    def _finish(self, found, result, stats):
        self.last_stats = stats.as_dict()
        if not found.value:
            return None
        return dict(score=float(result.score), pose_estimate=Rigid3d.from_c(result.pose_estimate),
                    rotational_score=float(result.rotational_score),
                    low_resolution_score=float(result.low_resolution_score))

    def match(self, global_node_pose, global_submap_pose, constant_data, min_score):
        node, submap = global_node_pose.to_c(), global_submap_pose.to_c()
        data = constant_data.to_c()
        found, result, stats = C.c_int32(), Result3D(), MatchStats()
        check(_lib.lib().cmx_fast3d_match(self._h, C.byref(node), C.byref(submap), C.byref(data),
                                          min_score, C.byref(found), C.byref(result),
                                          C.byref(stats)))
        return self._finish(found, result, stats)

    def match_full_submap(self, global_node_rotation, global_submap_rotation, constant_data,
                          min_score):
        nq = np.ascontiguousarray(global_node_rotation, np.float64)
        sq = np.ascontiguousarray(global_submap_rotation, np.float64)
        data = constant_data.to_c()
        found, result, stats = C.c_int32(), Result3D(), MatchStats()
        check(_lib.lib().cmx_fast3d_match_full_submap(self._h, nq.ctypes.data, sq.ctypes.data,
                                                      C.byref(data), min_score, C.byref(found),
                                                      C.byref(result), C.byref(stats)))
        return self._finish(found, result, stats)

    def level(self, depth):
        """Non-zero cells of one precomputation level, int32 [n,4] (x,y,z,value) sorted (z,y,x)."""
        lo = np.zeros(3, np.int32)
        dims = np.zeros(3, np.int32)
        check(_lib.lib().cmx_fast3d_level_info(self._h, depth, lo.ctypes.data, dims.ctypes.data))
        cells = np.empty((dims[2], dims[1], dims[0]), np.uint8)
        check(_lib.lib().cmx_fast3d_level_cells(self._h, depth, cells.ctypes.data))
        z, y, x = np.nonzero(cells)
        out = np.stack([x + lo[0], y + lo[1], z + lo[2], cells[z, y, x]], 1).astype(np.int32)
        return out


def fast3d_match_batch(matchers, node_poses, submap_poses, match_full_submap, min_scores,
                       constant_data):
    """One node's data against many submaps' matchers (ConstraintBuilder3D's fan-out,
    ``constraints/constraint_builder_3d.cc:79-147``) in one call: ``cmx_fast3d_match_batch`` runs
    the pairs concurrently on separate streams.

    ``node_poses[p]`` / ``submap_poses[p]`` are ``Rigid3d`` (only the rotations are read for
    full-submap pairs).  Returns (list of result dicts or None per pair, stats dict).
    """
    num = len(matchers)
    handles = (C.c_void_p * num)(*[m._h for m in matchers])
    nodes = (Pose3d * num)(*[p.to_c() for p in node_poses])
    submaps = (Pose3d * num)(*[p.to_c() for p in submap_poses])
    full = np.ascontiguousarray([1 if f else 0 for f in match_full_submap], np.int32)
    thresholds = np.ascontiguousarray(min_scores, np.float32)
    assert full.shape[0] == num and thresholds.shape[0] == num
    data = constant_data.to_c()
    found = np.zeros(num, np.int32)
    results = (Result3D * num)()
    stats = MatchStats()
    check(_lib.lib().cmx_fast3d_match_batch(
        handles, num, C.cast(nodes, C.c_void_p), C.cast(submaps, C.c_void_p), full.ctypes.data,
        thresholds.ctypes.data, C.byref(data), found.ctypes.data, C.cast(results, C.c_void_p),
        C.byref(stats)))
    out = []
    for p in range(num):
        if not found[p]:
            out.append(None)
            continue
        r = results[p]
        out.append(dict(score=float(r.score), pose_estimate=Rigid3d.from_c(r.pose_estimate),
                        rotational_score=float(r.rotational_score),
                        low_resolution_score=float(r.low_resolution_score)))
    return out, stats.as_dict()
