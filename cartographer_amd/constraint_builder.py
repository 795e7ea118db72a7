"""Host-side mirror of ``constraints::ConstraintBuilder2D`` with a batched device back end.

Reference: ``cartographer/mapping/internal/constraints/constraint_builder_2d.{h,cc}``.
Same entry points, argument meaning and filtering rules (``MaybeAddConstraint``:
``max_constraint_distance`` and the per-submap ``FixedRatioSampler``;
``MaybeAddGlobalConstraint``; ``NotifyEndOfNode``; ``WhenDone``; ``DeleteScanMatcher``;
``GetNumFinishedNodes``).  What differs is the execution model: the reference schedules one
thread-pool task per (node, submap) pair (:97-111); here the pairs of a node are collected and
``NotifyEndOfNode`` searches them in ONE device batch (``cmx_fast2d_match_batch``), against
precomputation stacks that stay resident in HBM, keyed by ``SubmapId``, until
``DeleteScanMatcher`` (:307-316) frees them.

Not built: the Ceres refinement of :240-249.  ``refine`` may be given a
callable ``(pose_estimate, point_cloud, grid) -> Rigid2d`` standing in for it; by default the
correlative estimate is used unrefined.
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import MatchStats, Pose2d, check
from .scan_matching import FastCorrelativeScanMatcher2D, Grid2D, Rigid2d, _cloud

SubmapId = Tuple[int, int]     # (trajectory_id, submap_index)
NodeId = Tuple[int, int]       # (trajectory_id, node_index)


def multiply(lhs: Rigid2d, rhs: Rigid2d) -> Rigid2d:
    """Rigid2d operator* (transform/rigid_transform.h:88-94)."""
    c, s = math.cos(lhs.theta), math.sin(lhs.theta)
    return Rigid2d(c * rhs.x - s * rhs.y + lhs.x, s * rhs.x + c * rhs.y + lhs.y,
                   lhs.theta + rhs.theta)


def inverse(t: Rigid2d) -> Rigid2d:
    """Rigid2d::inverse (transform/rigid_transform.h:73-77)."""
    c, s = math.cos(-t.theta), math.sin(-t.theta)
    return Rigid2d(-(c * t.x - s * t.y), -(s * t.x + c * t.y), -t.theta)


class FixedRatioSampler:
    """common::FixedRatioSampler (common/fixed_ratio_sampler.cc:24-39)."""

    def __init__(self, ratio: float):
        if not 0.0 <= ratio <= 1.0:
            raise ValueError("sampling ratio must be in [0, 1]")      # CHECK_GE / CHECK_LE
        self.ratio = ratio
        self.num_pulses = 0
        self.num_samples = 0

    def pulse(self) -> bool:
        self.num_pulses += 1
        if self.num_samples / self.num_pulses < self.ratio:
            self.num_samples += 1
            return True
        return False


@dataclass
class ConstraintBuilderOptions:
    """constraints::proto::ConstraintBuilderOptions; defaults of configuration_files/pose_graph.lua."""
    sampling_ratio: float = 0.3
    max_constraint_distance: float = 15.0
    min_score: float = 0.55
    global_localization_min_score: float = 0.6
    loop_closure_translation_weight: float = 1.1e4
    loop_closure_rotation_weight: float = 1e5
    linear_search_window: float = 7.0
    angular_search_window: float = math.radians(30.0)
    branch_and_bound_depth: int = 7


@dataclass
class Submap2D:
    """What ConstraintBuilder2D reads of a finished submap: its pose and its grid."""
    local_pose: Rigid2d          # ComputeSubmapPose(): Project2D(submap.local_pose())
    grid: Grid2D


@dataclass
class Constraint:
    """PoseGraphInterface::Constraint (tag is always INTER_SUBMAP here)."""
    submap_id: SubmapId
    node_id: NodeId
    zbar_ij: Rigid2d                      # submap i <- node j (Embed3D adds z = 0, rotation about z)
    translation_weight: float
    rotation_weight: float
    score: float
    tag: str = "INTER_SUBMAP"


@dataclass
class _Pending:
    slot: int
    submap_id: SubmapId
    submap: Submap2D
    node_id: NodeId
    point_cloud: np.ndarray
    cloud_key: int
    match_full_submap: bool
    initial_relative_pose: Rigid2d
    # The matcher the pair was scheduled on: DeleteScanMatcher only drops the builder's entry,
    # work already scheduled keeps its matcher alive (constraint_builder_2d.cc:307-316 frees the
    # matcher in a task that depends on the scheduled ones).
    matcher: object = None


class ConstraintBuilder2D:
    def __init__(self, options: ConstraintBuilderOptions, device: int = 0,
                 refine: Optional[Callable] = None, ceres=None):
        # `ceres`: a scan_matching.CeresScanMatcher2D, the builder's ceres_scan_matcher_; the
        # found pairs of a node are refined by one cmx_fast2d_refine_batch launch.
        self.options = options
        self.device = device
        self.refine = refine
        self.ceres = ceres
        self.last_refine_summaries = None
        self._scan_matchers: Dict[SubmapId, FastCorrelativeScanMatcher2D] = {}
        self._samplers: Dict[SubmapId, FixedRatioSampler] = {}
        self._constraints: List[Optional[Constraint]] = []     # constraints_ (:139-141 of the .h)
        self._pending: List[_Pending] = []
        self._num_started_nodes = 0
        self._num_finished_nodes = 0
        self.score_histogram: List[float] = []
        self.last_batch_stats = None

    # -- reference API -------------------------------------------------------
    def maybe_add_constraint(self, submap_id: SubmapId, submap: Submap2D, node_id: NodeId,
                             point_cloud, initial_relative_pose: Rigid2d) -> None:
        """MaybeAddConstraint (:77-111): windowed search around submap_pose * initial_relative_pose."""
        if math.hypot(initial_relative_pose.x, initial_relative_pose.y) > \
                self.options.max_constraint_distance:
            return
        sampler = self._samplers.setdefault(submap_id,
                                            FixedRatioSampler(self.options.sampling_ratio))
        if not sampler.pulse():
            return
        self._enqueue(submap_id, submap, node_id, point_cloud, False, initial_relative_pose)

    def maybe_add_global_constraint(self, submap_id: SubmapId, submap: Submap2D, node_id: NodeId,
                                    point_cloud) -> None:
        """MaybeAddGlobalConstraint (:113-137): full-submap search, no sampling, no distance cut."""
        self._enqueue(submap_id, submap, node_id, point_cloud, True, Rigid2d())

    def notify_end_of_node(self) -> None:
        """NotifyEndOfNode (:139-151).  The node's pairs are searched here, in one device batch."""
        self._num_started_nodes += 1
        self._flush()
        self._num_finished_nodes += 1

    def when_done(self, callback: Callable[[List[Constraint]], None]) -> None:
        """WhenDone / RunWhenDoneCallback (:153-163, :277-299): the non-null constraints in the
        order the pairs were added, then the queue is cleared."""
        self._flush()
        result = [c for c in self._constraints if c is not None]
        self._constraints = []
        callback(result)

    def get_num_finished_nodes(self) -> int:
        return self._num_finished_nodes

    def delete_scan_matcher(self, submap_id: SubmapId) -> None:
        """DeleteScanMatcher (:307-316): frees the submap's device stack and its sampler."""
        self._scan_matchers.pop(submap_id, None)
        self._samplers.pop(submap_id, None)

    def num_scan_matchers(self) -> int:
        return len(self._scan_matchers)

    # -- internals -------------------------------------------------------------
    def _enqueue(self, submap_id, submap, node_id, point_cloud, full, initial_relative_pose):
        self._constraints.append(None)
        # DispatchScanMatcherConstruction (:165-186): one matcher per submap id, built once.
        if submap_id not in self._scan_matchers:
            self._scan_matchers[submap_id] = FastCorrelativeScanMatcher2D(
                submap.grid, self.options.branch_and_bound_depth,
                self.options.linear_search_window, self.options.angular_search_window,
                device=self.device)
        # Pairs that share the caller's point-cloud object (one node's constant data) go
        # into the same device batch.
        xyz, _ = _cloud(point_cloud)
        self._pending.append(_Pending(len(self._constraints) - 1, submap_id, submap, node_id, xyz,
                                      id(point_cloud), full, initial_relative_pose,
                                      self._scan_matchers[submap_id]))

    def _flush(self):
        pending, self._pending = self._pending, []
        groups: Dict[int, List[_Pending]] = {}
        for item in pending:                      # one batch per point cloud (= per node)
            groups.setdefault(item.cloud_key, []).append(item)
        for items in groups.values():
            self._match_group(items[0].point_cloud, items)

    def _match_group(self, xyz, items):
        num = len(items)
        handles = (C.c_void_p * num)(*[i.matcher._h for i in items])
        initial = (Pose2d * num)()
        full = np.zeros(num, np.int32)
        min_scores = np.zeros(num, np.float32)
        for k, item in enumerate(items):
            # ComputeConstraint (:196-197): initial_pose = ComputeSubmapPose(submap) * relative.
            init = multiply(item.submap.local_pose, item.initial_relative_pose)
            initial[k] = Pose2d(init.x, init.y, init.theta)
            full[k] = 1 if item.match_full_submap else 0
            min_scores[k] = (self.options.global_localization_min_score if item.match_full_submap
                             else self.options.min_score)
        found = np.zeros(num, np.int32)
        scores = np.zeros(num, np.float32)
        poses = (Pose2d * num)()
        stats = MatchStats()
        check(_lib.lib().cmx_fast2d_match_batch(
            handles, num, C.cast(initial, C.c_void_p), full.ctypes.data, min_scores.ctypes.data,
            xyz.ctypes.data, xyz.shape[0], found.ctypes.data, scores.ctypes.data,
            C.cast(poses, C.c_void_p), C.byref(stats)))
        self.last_batch_stats = stats.as_dict()
        refined = None
        if self.ceres is not None and found.any():
            refined, self.last_refine_summaries = self.ceres.refine_batch(
                [i.matcher for i in items], found,
                [Rigid2d(p.x, p.y, p.theta) for p in poses], xyz)
        for k, item in enumerate(items):
            if not found[k]:
                continue                                   # `return;` at :219 / :232
            score = float(scores[k])
            self.score_histogram.append(score)
            pose_estimate = Rigid2d(poses[k].x, poses[k].y, poses[k].theta)
            if refined is not None:                        # ceres_scan_matcher_.Match (:245-249)
                pose_estimate = refined[k]
            elif self.refine is not None:                  # host callback (experiments)
                pose_estimate = self.refine(pose_estimate, item.point_cloud, item.submap.grid)
            constraint_transform = multiply(inverse(item.submap.local_pose), pose_estimate)
            self._constraints[item.slot] = Constraint(
                item.submap_id, item.node_id, constraint_transform,
                self.options.loop_closure_translation_weight,
                self.options.loop_closure_rotation_weight, score)


# ---------------------------------------------------------------------------- 3D
@dataclass
class ConstraintBuilderOptions3D:
    """constraints::proto::ConstraintBuilderOptions with its 3D matcher options; defaults of
    configuration_files/pose_graph.lua."""
    sampling_ratio: float = 0.3
    max_constraint_distance: float = 15.0
    min_score: float = 0.55
    global_localization_min_score: float = 0.6
    loop_closure_translation_weight: float = 1.1e4
    loop_closure_rotation_weight: float = 1e5
    branch_and_bound_depth: int = 8
    full_resolution_depth: int = 3
    min_rotational_score: float = 0.77
    min_low_resolution_score: float = 0.55
    linear_xy_search_window: float = 5.0
    linear_z_search_window: float = 1.0
    angular_search_window: float = math.radians(15.0)


@dataclass
class Submap3D:
    """What ConstraintBuilder3D reads of a finished submap (constraint_builder_3d.cc:176-186):
    both hybrid grids (flattened voxel lists, see scan_matching_3d) and the histogram."""
    high_resolution: float
    high_resolution_voxels: np.ndarray
    grid_size: int
    low_resolution: float
    low_resolution_voxels: np.ndarray
    rotational_scan_matcher_histogram: np.ndarray


@dataclass
class Constraint3D:
    submap_id: SubmapId
    node_id: NodeId
    zbar_ij: object                       # scan_matching_3d.Rigid3d, submap i <- node j
    translation_weight: float
    rotation_weight: float
    score: float
    rotational_score: float
    low_resolution_score: float
    tag: str = "INTER_SUBMAP"


class ConstraintBuilder3D:
    """Host-side mirror of ``constraints::ConstraintBuilder3D``
    (``cartographer/mapping/internal/constraints/constraint_builder_3d.{h,cc}``): the distance
    filter on the GLOBAL poses (:84-87), the per-submap ``FixedRatioSampler`` (:88-93), one
    matcher per ``SubmapId`` resident in HBM until ``DeleteScanMatcher``, ``Match`` /
    ``MatchFullSubmap`` with the two thresholds (:218-255), results in the order the pairs were
    added (``RunWhenDoneCallback``).  ``NotifyEndOfNode`` hands the node's pairs to
    ``cmx_fast3d_match_batch`` (one chain of launches for all pairs) and, with ``ceres`` (a
    ``scan_matching_3d.CeresScanMatcher3D`` with two occupied-space weights: the builder's
    ``ceres_scan_matcher_``), the found ones to ``cmx_fast3d_refine_batch`` (:263-276) -- both
    against grids that stay in HBM.  ``refine`` is a host callback instead, for experiments.
    """

    def __init__(self, options: ConstraintBuilderOptions3D, device: int = 0,
                 refine: Optional[Callable] = None, ceres=None):
        self.options = options
        self.device = device
        self.refine = refine
        self.ceres = ceres
        self.last_refine_summaries = None
        self._scan_matchers = {}
        self._samplers: Dict[SubmapId, FixedRatioSampler] = {}
        self._constraints: List[Optional[Constraint3D]] = []
        self._pending = []
        self._num_finished_nodes = 0
        self.score_histogram: List[float] = []
        self.last_batch_stats = None

    def maybe_add_constraint(self, submap_id, submap: Submap3D, node_id, constant_data,
                             global_node_pose, global_submap_pose) -> None:
        delta = (np.asarray(global_node_pose.translation, np.float64) -
                 np.asarray(global_submap_pose.translation, np.float64))
        # Eigen's fixed-size norm: sqrt((x*x + y*y) + z*z) in f64
        if math.sqrt((delta[0] * delta[0] + delta[1] * delta[1]) + delta[2] * delta[2]) > \
                self.options.max_constraint_distance:
            return
        sampler = self._samplers.setdefault(submap_id,
                                            FixedRatioSampler(self.options.sampling_ratio))
        if not sampler.pulse():
            return
        self._enqueue(submap_id, submap, node_id, constant_data, False, global_node_pose,
                      global_submap_pose)

    def maybe_add_global_constraint(self, submap_id, submap: Submap3D, node_id, constant_data,
                                    global_node_rotation, global_submap_rotation) -> None:
        self._enqueue(submap_id, submap, node_id, constant_data, True, global_node_rotation,
                      global_submap_rotation)

    def notify_end_of_node(self) -> None:
        self._flush()
        self._num_finished_nodes += 1

    def when_done(self, callback) -> None:
        self._flush()
        result = [c for c in self._constraints if c is not None]
        self._constraints = []
        callback(result)

    def get_num_finished_nodes(self) -> int:
        return self._num_finished_nodes

    def delete_scan_matcher(self, submap_id) -> None:
        self._scan_matchers.pop(submap_id, None)
        self._samplers.pop(submap_id, None)

    def num_scan_matchers(self) -> int:
        return len(self._scan_matchers)

    def _enqueue(self, submap_id, submap, node_id, constant_data, full, node, sub):
        from .scan_matching_3d import FastCorrelativeScanMatcher3D
        self._constraints.append(None)
        if submap_id not in self._scan_matchers:          # DispatchScanMatcherConstruction
            o = self.options
            self._scan_matchers[submap_id] = FastCorrelativeScanMatcher3D(
                submap.high_resolution, submap.high_resolution_voxels, submap.grid_size,
                submap.low_resolution, submap.low_resolution_voxels,
                submap.rotational_scan_matcher_histogram,
                branch_and_bound_depth=o.branch_and_bound_depth,
                full_resolution_depth=o.full_resolution_depth,
                min_rotational_score=o.min_rotational_score,
                min_low_resolution_score=o.min_low_resolution_score,
                linear_xy_search_window=o.linear_xy_search_window,
                linear_z_search_window=o.linear_z_search_window,
                angular_search_window=o.angular_search_window, device=self.device)
        # (the matcher travels with the pair: a later delete_scan_matcher must not break it)
        self._pending.append((len(self._constraints) - 1, submap_id, node_id, constant_data, full,
                              node, sub, self._scan_matchers[submap_id]))

    def _flush(self):
        """The queued pairs of a node share its constant data: one cmx_fast3d_match_batch per
        node (pairs searched concurrently on the device), in the order they were added."""
        from .scan_matching_3d import Rigid3d, fast3d_match_batch
        pending, self._pending = self._pending, []
        groups: Dict[int, list] = {}
        for item in pending:
            groups.setdefault(id(item[3]), []).append(item)
        for items in groups.values():
            constant_data = items[0][3]
            as_pose = lambda v, full: Rigid3d((0.0, 0.0, 0.0), tuple(v)) if full else v   # noqa: E731
            results, self.last_batch_stats = fast3d_match_batch(
                [i[7] for i in items],
                [as_pose(i[5], i[4]) for i in items], [as_pose(i[6], i[4]) for i in items],
                [i[4] for i in items],
                [self.options.global_localization_min_score if i[4] else self.options.min_score
                 for i in items], constant_data)
            refined = None
            if self.ceres is not None and any(r is not None for r in results):
                identity = Rigid3d()
                refined, self.last_refine_summaries = self.ceres.refine_batch(
                    [i[7] for i in items], [r is not None for r in results],
                    [r["pose_estimate"] if r is not None else identity for r in results],
                    constant_data)
            for k, ((slot, submap_id, node_id, _, full, node, sub, _m), result) in enumerate(
                    zip(items, results)):
                if result is None:
                    continue                               # `return;` at :232 / :253
                self.score_histogram.append(result["score"])
                pose = result["pose_estimate"]             # already submap i <- node j
                if refined is not None:
                    pose = refined[k]                      # ceres_scan_matcher_.Match (:263-276)
                elif self.refine is not None:
                    pose = self.refine(pose, constant_data)
                self._constraints[slot] = Constraint3D(
                    submap_id, node_id, pose, self.options.loop_closure_translation_weight,
                    self.options.loop_closure_rotation_weight, result["score"],
                    result["rotational_score"], result["low_resolution_score"])
