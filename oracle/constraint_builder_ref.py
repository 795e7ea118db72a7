"""CPU restatement of constraints::ConstraintBuilder2D (test infrastructure only).

Follows cartographer/mapping/internal/constraints/constraint_builder_2d.cc:77-137 (filters and
queueing), :188-262 (ComputeConstraint without the Ceres refinement of :245-249, which is outside
this repository's scope) and :277-299 (result order), one (node, submap) pair at a time through
the oracle's FastCorrelativeScanMatcher2D, exactly as the reference's tasks run them.
Rigid2d algebra: transform/rigid_transform.h:73-94; sampler: common/fixed_ratio_sampler.cc:24-39.
"""
import math

from . import pyoracle as orc


def rigid_mul(a, b):
    c, s = math.cos(a[2]), math.sin(a[2])
    return (c * b[0] - s * b[1] + a[0], s * b[0] + c * b[1] + a[1], a[2] + b[2])


def rigid_inv(t):
    c, s = math.cos(-t[2]), math.sin(-t[2])
    return (-(c * t[0] - s * t[1]), -(s * t[0] + c * t[1]), -t[2])


class ConstraintBuilder2DRef:
    def __init__(self, sampling_ratio, max_constraint_distance, min_score,
                 global_localization_min_score, linear_search_window, angular_search_window,
                 branch_and_bound_depth):
        self.o = dict(sampling_ratio=sampling_ratio, max_constraint_distance=max_constraint_distance,
                      min_score=min_score,
                      global_localization_min_score=global_localization_min_score,
                      lin=linear_search_window, ang=angular_search_window,
                      depth=branch_and_bound_depth)
        self.matchers = {}
        self.samplers = {}        # submap_id -> [num_pulses, num_samples]
        self.constraints = []
        self.finished = 0

    def _matcher(self, submap_id, grid):
        if submap_id not in self.matchers:
            cells, res, max_x, max_y = grid
            self.matchers[submap_id] = orc.FastCorrelativeScanMatcher2D(
                cells, res, max_x, max_y, self.o["depth"], self.o["lin"], self.o["ang"])
        return self.matchers[submap_id]

    def maybe_add_constraint(self, submap_id, submap_pose, grid, node_id, cloud, rel):
        if math.hypot(rel[0], rel[1]) > self.o["max_constraint_distance"]:
            return
        st = self.samplers.setdefault(submap_id, [0, 0])
        st[0] += 1
        if not (st[1] / st[0] < self.o["sampling_ratio"]):
            return
        st[1] += 1
        self._compute(submap_id, submap_pose, grid, node_id, cloud, False, rel)

    def maybe_add_global_constraint(self, submap_id, submap_pose, grid, node_id, cloud):
        self._compute(submap_id, submap_pose, grid, node_id, cloud, True, (0.0, 0.0, 0.0))

    def _compute(self, submap_id, submap_pose, grid, node_id, cloud, full, rel):
        m = self._matcher(submap_id, grid)
        if full:
            r = m.match_full_submap(cloud, self.o["global_localization_min_score"])
        else:
            r = m.match(list(rigid_mul(submap_pose, rel)), cloud, self.o["min_score"])
        if not r["found"]:
            self.constraints.append(None)
            return
        pose = tuple(r["pose"])
        self.constraints.append(dict(submap_id=submap_id, node_id=node_id,
                                     zbar_ij=rigid_mul(rigid_inv(submap_pose), pose),
                                     score=r["score"]))

    def notify_end_of_node(self):
        self.finished += 1

    def when_done(self):
        out = [c for c in self.constraints if c is not None]
        self.constraints = []
        return out

    def delete_scan_matcher(self, submap_id):
        self.matchers.pop(submap_id, None)
        self.samplers.pop(submap_id, None)


class ConstraintBuilder3DRef:
    """CPU restatement of constraints::ConstraintBuilder3D
    (cartographer/mapping/internal/constraints/constraint_builder_3d.cc:79-147 filters and
    queueing, :199-283 ComputeConstraint without the Ceres refinement of :263-276), one pair at a
    time through the oracle's FastCorrelativeScanMatcher3D.  Poses are 7-vectors
    (t xyz, q wxyz); `submap` = (resolution, voxels, low_resolution, low_voxels, histogram);
    `data` = (gravity wxyz, high-resolution cloud, low-resolution cloud, histogram)."""

    def __init__(self, sampling_ratio, max_constraint_distance, min_score,
                 global_localization_min_score, depth, full_resolution_depth,
                 min_rotational_score, min_low_resolution_score, lin_xy, lin_z, ang):
        self.o = dict(sampling_ratio=sampling_ratio, max_constraint_distance=max_constraint_distance,
                      min_score=min_score,
                      global_localization_min_score=global_localization_min_score)
        self.matcher_args = (depth, full_resolution_depth, min_rotational_score,
                             min_low_resolution_score, lin_xy, lin_z, ang)
        self.matchers = {}
        self.samplers = {}
        self.constraints = []
        self.finished = 0

    def _matcher(self, submap_id, submap):
        if submap_id not in self.matchers:
            res, vox, low_res, low_vox, hist = submap
            self.matchers[submap_id] = orc.FastCorrelativeScanMatcher3D(
                res, vox, low_res, low_vox, hist, *self.matcher_args)
        return self.matchers[submap_id]

    def maybe_add_constraint(self, submap_id, submap, node_id, data, node_pose, submap_pose):
        d = [node_pose[k] - submap_pose[k] for k in range(3)]
        if math.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) > \
                self.o["max_constraint_distance"]:
            return
        st = self.samplers.setdefault(submap_id, [0, 0])
        st[0] += 1
        if not (st[1] / st[0] < self.o["sampling_ratio"]):
            return
        st[1] += 1
        self._compute(submap_id, submap, node_id, data, False, node_pose, submap_pose)

    def maybe_add_global_constraint(self, submap_id, submap, node_id, data, node_q, submap_q):
        self._compute(submap_id, submap, node_id, data, True, [0, 0, 0] + list(node_q),
                      [0, 0, 0] + list(submap_q))

    def _compute(self, submap_id, submap, node_id, data, full, node, sub):
        m = self._matcher(submap_id, submap)
        gravity, hi, lo, hist = data
        if full:
            r = m.match_full_submap(node[3:], sub[3:], gravity, hi, lo, hist,
                                    self.o["global_localization_min_score"])
        else:
            r = m.match(node, sub, gravity, hi, lo, hist, self.o["min_score"])
        if not r["found"]:
            self.constraints.append(None)
            return
        self.constraints.append(dict(submap_id=submap_id, node_id=node_id,
                                     zbar_ij=list(r["pose"]), score=r["score"],
                                     rotational_score=r["rotational_score"],
                                     low_resolution_score=r["low_resolution_score"]))

    def notify_end_of_node(self):
        self.finished += 1

    def when_done(self):
        out = [c for c in self.constraints if c is not None]
        self.constraints = []
        return out

    def delete_scan_matcher(self, submap_id):
        self.matchers.pop(submap_id, None)
        self.samplers.pop(submap_id, None)
