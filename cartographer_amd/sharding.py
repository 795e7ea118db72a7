"""Multi-GPU sharding of the loop-closure search (one scan vs many submaps).

Every (scan, submap) search is independent
(``cartographer/mapping/internal/constraints/constraint_builder_2d.cc:97-111`` schedules them as
independent tasks), so submaps are partitioned over the ranks with no data-path collective.  The
only exchange is the node-wide best match: ONE all-reduce(max) of an 8-byte key
``score_bits << 32 | (0xFFFFFFFF - global_submap_id)`` (positive f32 bit patterns order like the
floats; equal scores resolve to the lowest submap id on any number of ranks).
``torch.distributed`` backend "nccl" is RCCL over xGMI on MI355X; the same code runs on "gloo".
"""
import numpy as np


def shard_range(num_items, rank, world_size):
    """Contiguous block partition: items [begin, end) of rank `rank`."""
    base, extra = divmod(num_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


NOT_FOUND = -1       # sentinel key: no rank found a match (distinct from every real key, all >= 0)


def pack_best_key(found, scores, first_global_id):
    """Key of the best local match: ``score_bits << 32 | (0xFFFFFFFF - global id)``, so that the
    MAX over ranks picks the best score and, among equal scores, the LOWEST submap id -- the same
    rule as within a rank (first maximum), whatever the sharding.  NOT_FOUND when nothing was
    found."""
    found = np.asarray(found)
    scores = np.asarray(scores, np.float32)
    if found.size == 0 or not found.any():
        return NOT_FOUND
    masked = np.where(found > 0, scores, np.float32(-1.0))
    i = int(np.argmax(masked))            # first maximum: lowest submap id wins ties locally
    bits = int(scores[i:i + 1].view(np.uint32)[0])      # scores > 0: sign bit clear, key >= 0
    return (bits << 32) | (0xFFFFFFFF - (first_global_id + i))


def unpack_best_key(key):
    """Returns (score or None, global submap id or None)."""
    key = int(key)
    if key < 0:
        return None, None
    score = float(np.array([key >> 32], np.uint32).view(np.float32)[0])
    return score, 0xFFFFFFFF - (key & 0xFFFFFFFF)


def all_reduce_best(key, device=None):
    """MAX all-reduce of the packed key across the default process group."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([key], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


RESULT_WORDS = 6   # found, score, x, y, theta, (pad) as float64 per submap


def all_gather_results(found, scores, poses_xyt, num_submaps, rank, world_size, device=None):
    """Every rank learns every submap's optional constraint: the reference semantics (one
    optional constraint per (node, submap) pair collected by WhenDone,
    constraint_builder_2d.cc:277-299) across ranks.  Each rank contributes the results of its
    `shard_range` block; the exchange is one all-gather of RESULT_WORDS float64 per submap
    (512 submaps: 24 KB -- latency-bound on xGMI, like the all-reduce above).

    Returns (found[num_submaps], scores[num_submaps] f32, poses[num_submaps, 3] f64) in global
    submap order.
    """
    import torch
    import torch.distributed as dist
    begin, end = shard_range(num_submaps, rank, world_size)
    assert len(found) == end - begin
    per_rank = -(-num_submaps // world_size)                     # equal-sized slots
    local = np.zeros((per_rank, RESULT_WORDS), np.float64)
    local[:end - begin, 0] = np.asarray(found, np.float64)
    local[:end - begin, 1] = np.asarray(scores, np.float32).astype(np.float64)   # exact widening
    local[:end - begin, 2:5] = np.asarray(poses_xyt, np.float64).reshape(-1, 3)
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and world_size > 1:
        gathered = [torch.empty_like(t) for _ in range(world_size)]
        dist.all_gather(gathered, t)
    else:
        gathered = [t]
    out = np.zeros((num_submaps, RESULT_WORDS), np.float64)
    for r, g in enumerate(gathered):
        b, e = shard_range(num_submaps, r, world_size)
        out[b:e] = g.cpu().numpy()[:e - b]
    return (out[:, 0].astype(np.int32), out[:, 1].astype(np.float32), out[:, 2:5].copy())


def all_gather_rows(rows, num_items, rank, world_size, device=None):
    """The generic form of all_gather_results: `rows` is this rank's [end - begin, width] float64
    block (shard_range order); returns the [num_items, width] array every rank ends up with.
    Used for 3D results (found, score, translation, quaternion = 9 words per submap)."""
    import torch
    import torch.distributed as dist
    rows = np.ascontiguousarray(rows, np.float64)
    begin, end = shard_range(num_items, rank, world_size)
    assert rows.shape[0] == end - begin
    width = rows.shape[1]
    per_rank = -(-num_items // world_size)
    local = np.zeros((per_rank, width), np.float64)
    local[:end - begin] = rows
    t = torch.from_numpy(local)
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and world_size > 1:
        gathered = [torch.empty_like(t) for _ in range(world_size)]
        dist.all_gather(gathered, t)
    else:
        gathered = [t]
    out = np.zeros((num_items, width), np.float64)
    for r, g in enumerate(gathered):
        b, e = shard_range(num_items, r, world_size)
        out[b:e] = g.cpu().numpy()[:e - b]
    return out


class Communicator:
    """cmx_comm: one host process driving several GPUs of the node (RCCL through
    ncclCommInitAll).  `match_full_submap_batch` / `match_batch` are the sharded forms of the
    functions of the same name in scan_matching: the matchers may live on different devices."""

    def __init__(self, devices):
        import ctypes as C
        from . import _lib
        self._h = C.c_void_p()
        devs = np.ascontiguousarray(devices, np.int32)
        _lib.check(_lib.lib().cmx_comm_init(devs.ctypes.data, len(devs), C.byref(self._h)))
        self.devices = list(map(int, devs))

    @property
    def num_devices(self):
        """Ranks of the communicator (= devices, unless the debug switch comm_virtual_ranks made
        one device several ranks)."""
        from . import _lib
        return int(_lib.lib().cmx_comm_num_devices(self._h))

    @property
    def uses_rccl(self):
        """True if the best-match key is reduced by RCCL (several devices; one with the debug
        switch comm_force_rccl), False if the communicator needs no collective."""
        from . import _lib
        return bool(_lib.lib().cmx_comm_uses_rccl(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                from . import _lib
                _lib.lib().cmx_comm_destroy(self._h)
            except ImportError:      # interpreter shutting down: the process frees everything
                pass
            self._h = None

    def device_of(self, index, num_items):
        from . import _lib
        return _lib.lib().cmx_comm_device_of(self._h, index, num_items)

    def match_batch(self, matchers, initial_pose_estimates, match_full_submap, min_scores,
                    point_cloud):
        """Returns (found, scores, poses, best (index or -1, score), stats)."""
        import ctypes as C
        from . import _lib, scan_matching as sm
        num = len(matchers)
        handles = (C.c_void_p * num)(*[m._h for m in matchers])
        initial = (_lib.Pose2d * num)(*[p.to_c() for p in initial_pose_estimates])
        full = np.ascontiguousarray(match_full_submap, np.int32)
        thresholds = np.ascontiguousarray(min_scores, np.float32)
        xyz = np.ascontiguousarray(point_cloud, np.float32).reshape(-1, 3)
        found = np.zeros(num, np.int32)
        scores = np.zeros(num, np.float32)
        poses = (_lib.Pose2d * num)()
        best_index, best_score, stats = C.c_int32(), C.c_float(), _lib.MatchStats()
        _lib.check(_lib.lib().cmx_fast2d_match_sharded(
            self._h, handles, num, C.cast(initial, C.c_void_p), full.ctypes.data,
            thresholds.ctypes.data, xyz.ctypes.data, xyz.shape[0], found.ctypes.data,
            scores.ctypes.data, C.cast(poses, C.c_void_p), C.byref(best_index),
            C.byref(best_score), C.byref(stats)))
        return (found, scores, [sm.Rigid2d(p.x, p.y, p.theta) for p in poses],
                (best_index.value, best_score.value), stats.as_dict())

    def match_batch_3d(self, matchers, node_poses, submap_poses, match_full_submap, min_scores,
                       constant_data):
        """cmx_fast3d_match_sharded: ConstraintBuilder3D's fan-out for one node
        (constraint_builder_3d.cc:79-147) with the matchers spread over the communicator's
        devices (BASELINE config C5: 256 submaps on 8 GPUs).  Returns (list of result dicts or
        None per pair, best (index or -1, score), stats)."""
        import ctypes as C
        from . import _lib, scan_matching_3d as sm3
        num = len(matchers)
        handles = (C.c_void_p * num)(*[m._h for m in matchers])
        nodes = (_lib.Pose3d * num)(*[p.to_c() for p in node_poses])
        submaps = (_lib.Pose3d * num)(*[p.to_c() for p in submap_poses])
        full = np.ascontiguousarray([1 if f else 0 for f in match_full_submap], np.int32)
        thresholds = np.ascontiguousarray(min_scores, np.float32)
        assert full.shape[0] == num and thresholds.shape[0] == num
        data = constant_data.to_c()
        found = np.zeros(num, np.int32)
        results = (_lib.Result3D * num)()
        best_index, best_score, stats = C.c_int32(), C.c_float(), _lib.MatchStats()
        _lib.check(_lib.lib().cmx_fast3d_match_sharded(
            self._h, C.cast(handles, C.c_void_p), num, C.cast(nodes, C.c_void_p),
            C.cast(submaps, C.c_void_p), full.ctypes.data, thresholds.ctypes.data, C.byref(data),
            found.ctypes.data, C.cast(results, C.c_void_p), C.byref(best_index),
            C.byref(best_score), C.byref(stats)))
        out = []
        for p in range(num):
            if not found[p]:
                out.append(None)
                continue
            r = results[p]
            out.append(dict(score=float(r.score),
                            pose_estimate=sm3.Rigid3d.from_c(r.pose_estimate),
                            rotational_score=float(r.rotational_score),
                            low_resolution_score=float(r.low_resolution_score)))
        return out, (best_index.value, best_score.value), stats.as_dict()
