"""Synthetic worlds, scans and probability grids (host-only tooling).

ctypes binding of ``cartographer_amd/lib/libcmx_synth.so`` (sources in
``csrc/host``): the probability-grid range-data inserter restatement
(reference ``mapping/2d/probability_grid_range_data_inserter_2d.cc:35-133``)
and the seeded room / lidar generator SURVEY.md §8d describes.  Used by
``bench.py`` and the tests to produce identical bytes for the GPU path and the
checker; not part of the device hot path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libcmx_synth.so")

_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"{_SO} is missing - run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_SO)
        L.cmx_synth_world_create.argtypes = [C.c_uint64] + [C.c_double] * 4
        L.cmx_synth_world_create.restype = C.c_void_p
        L.cmx_synth_world_destroy.argtypes = [C.c_void_p]
        L.cmx_synth_world_free_pose.argtypes = [C.c_void_p, C.c_uint64, C.c_double, _f64p]
        L.cmx_synth_scan.argtypes = [C.c_void_p, _f64p, C.c_int, C.c_double, C.c_double,
                                     C.c_uint64, _f32p]
        L.cmx_pgrid_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.cmx_pgrid_create.restype = C.c_void_p
        L.cmx_pgrid_destroy.argtypes = [C.c_void_p]
        L.cmx_pgrid_limits.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
        L.cmx_pgrid_cells.argtypes = [C.c_void_p, _u16p]
        L.cmx_pgrid_set_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.cmx_pgrid_get_probability.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cmx_pgrid_get_probability.restype = C.c_float
        L.cmx_pgrid_insert.argtypes = [C.c_void_p, _f32p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int, C.c_float, C.c_float, C.c_int]
        L.cmx_pgrid_cropped.argtypes = [C.c_void_p]
        L.cmx_pgrid_cropped.restype = C.c_void_p
        L.cmx_pgrid_odds_table.argtypes = [C.c_float, _u16p]
        L.cmx_cells_on_ray.argtypes = [C.c_int] * 5 + [_i32p, C.c_int, C.POINTER(C.c_int)]
        L.cmx_synth_submap.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_int,
                                       C.c_int, C.c_double, C.c_double, _u16p, _f64p]
        L.cmx_synth_submap.restype = C.c_void_p
        L.cmx_thread_driver_fast2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                               C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.cmx_thread_driver_fast2d.restype = C.c_double
        _lib = L
    return _lib


class ProbabilityGrid:
    """Mirror of mapping/2d/probability_grid.h for building fixtures."""

    def __init__(self, resolution, max_xy, num_x_cells, num_y_cells, _handle=None):
        self._h = _handle or lib().cmx_pgrid_create(resolution, max_xy[0], max_xy[1],
                                                    num_x_cells, num_y_cells)
        if not self._h:
            raise ValueError("invalid grid limits")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cmx_pgrid_destroy(self._h)
            self._h = None

    @property
    def limits(self):
        res, mx, my, nx, ny = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
        lib().cmx_pgrid_limits(self._h, C.byref(res), C.byref(mx), C.byref(my), C.byref(nx),
                               C.byref(ny))
        return dict(resolution=res.value, max_x=mx.value, max_y=my.value,
                    num_x_cells=nx.value, num_y_cells=ny.value)

    @property
    def cells(self):
        lim = self.limits
        out = np.empty((lim["num_y_cells"], lim["num_x_cells"]), np.uint16)
        lib().cmx_pgrid_cells(self._h, out)
        return out

    def set_probability(self, ix, iy, probability):
        if lib().cmx_pgrid_set_probability(self._h, ix, iy, probability):
            raise ValueError("SetProbability failed (cell known or outside)")

    def get_probability(self, ix, iy):
        return float(lib().cmx_pgrid_get_probability(self._h, ix, iy))

    def insert(self, origin_xy, returns_xyz, misses_xyz=None, hit_probability=0.7,
               miss_probability=0.4, insert_free_space=True):
        origin = np.ascontiguousarray(origin_xy, np.float32)[:2].copy()
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        mis = (np.ascontiguousarray(misses_xyz, np.float32).reshape(-1, 3)
               if misses_xyz is not None else np.zeros((0, 3), np.float32))
        rc = lib().cmx_pgrid_insert(self._h, origin, ret.ctypes.data, ret.shape[0],
                                    mis.ctypes.data, mis.shape[0], hit_probability,
                                    miss_probability, int(insert_free_space))
        if rc:
            raise RuntimeError("range data insertion failed")

    def cropped(self):
        return ProbabilityGrid(0, (0, 0), 0, 0, _handle=lib().cmx_pgrid_cropped(self._h))


def odds_table(probability):
    out = np.empty(32768, np.uint16)
    lib().cmx_pgrid_odds_table(probability, out)
    return out


def cells_on_ray(begin, end, scale=1000):
    cap = 1 << 16
    out = np.empty((cap, 2), np.int32)
    n = C.c_int()
    lib().cmx_cells_on_ray(begin[0], begin[1], end[0], end[1], scale, out, cap, C.byref(n))
    return out[: n.value].copy()


class World:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().cmx_synth_world_destroy(self._h)
            self._h = None

    def free_pose(self, seed, clearance=0.3):
        p = np.empty(3, np.float64)
        lib().cmx_synth_world_free_pose(self._h, seed, clearance, p)
        return p

    def scan(self, pose_xyt, beams=1000, max_range=10.0, sigma=0.01, seed=0):
        out = np.empty((beams, 3), np.float32)
        n = lib().cmx_synth_scan(self._h, np.ascontiguousarray(pose_xyt, np.float64), beams,
                                 max_range, sigma, seed, out)
        return out[:n].copy()


def make_submap(seed, nx=400, ny=400, resolution=0.05, num_poses=30, beams=1000,
                max_range=10.0, sigma=0.01):
    """Returns (cells[ny,nx] u16, limits dict, World)."""
    cells = np.empty((ny, nx), np.uint16)
    mx = np.empty(2, np.float64)
    h = lib().cmx_synth_submap(seed, nx, ny, resolution, num_poses, beams, max_range, sigma,
                               cells, mx)
    if not h:
        raise RuntimeError("synthetic submap generation failed (grid grew)")
    limits = dict(resolution=resolution, max_x=float(mx[0]), max_y=float(mx[1]),
                  num_x_cells=nx, num_y_cells=ny)
    return cells, limits, World(h)


# ---------------------------------------------------------------- 3D -------
VOXEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32), ("value", np.uint16),
                        ("pad", np.uint16)])


def _lib3d():
    L = lib()
    if not getattr(L, "_cmx_3d_declared", False):
        L.cmx_synth3d_world_create.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double]
        L.cmx_synth3d_world_create.restype = C.c_void_p
        L.cmx_synth3d_world_destroy.argtypes = [C.c_void_p]
        L.cmx_synth3d_free_position.argtypes = [C.c_void_p, C.c_uint64, C.c_double, _f64p]
        L.cmx_synth3d_scan.argtypes = [C.c_void_p, _f64p, C.c_double, C.c_int, C.c_int,
                                       C.c_double, C.c_double, C.c_double, C.c_uint64, _f32p]
        L.cmx_hgrid_create.argtypes = [C.c_float]
        L.cmx_hgrid_create.restype = C.c_void_p
        L.cmx_hgrid_destroy.argtypes = [C.c_void_p]
        L.cmx_hgrid_size.argtypes = [C.c_void_p]
        L.cmx_hgrid_set_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.cmx_hgrid_get_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.cmx_hgrid_get_probability.restype = C.c_float
        L.cmx_hgrid_cell_index.argtypes = [C.c_void_p, _f32p, _i32p]
        L.cmx_hgrid_insert.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_float, C.c_float,
                                       C.c_int]
        L.cmx_hgrid_num_voxels.argtypes = [C.c_void_p]
        L.cmx_hgrid_num_voxels.restype = C.c_int64
        L.cmx_hgrid_voxels.argtypes = [C.c_void_p, C.c_void_p]
        L._cmx_3d_declared = True
    return L


class HybridGrid:
    """Mirror of mapping/3d/hybrid_grid.h (write side) for building fixtures."""

    def __init__(self, resolution):
        self.resolution = float(np.float32(resolution))
        self._h = _lib3d().cmx_hgrid_create(resolution)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib3d().cmx_hgrid_destroy(self._h)
            self._h = None

    @property
    def grid_size(self):
        return int(_lib3d().cmx_hgrid_size(self._h))

    def get_cell_index(self, point):
        out = np.empty(3, np.int32)
        _lib3d().cmx_hgrid_cell_index(self._h, np.ascontiguousarray(point, np.float32), out)
        return out

    def set_probability(self, index, probability):
        _lib3d().cmx_hgrid_set_probability(self._h, int(index[0]), int(index[1]), int(index[2]),
                                           probability)

    def get_probability(self, index):
        return float(_lib3d().cmx_hgrid_get_probability(self._h, int(index[0]), int(index[1]),
                                                        int(index[2])))

    def insert(self, origin_xyz, returns_xyz, hit_probability=0.7, miss_probability=0.4,
               num_free_space_voxels=5):
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        _lib3d().cmx_hgrid_insert(self._h, np.ascontiguousarray(origin_xyz, np.float32), ret,
                                  ret.shape[0], hit_probability, miss_probability,
                                  num_free_space_voxels)

    def voxels(self):
        n = _lib3d().cmx_hgrid_num_voxels(self._h)
        out = np.zeros(n, VOXEL_DTYPE)
        if n:
            _lib3d().cmx_hgrid_voxels(self._h, out.ctypes.data)
        return out


class World3D:
    def __init__(self, seed, size=(15.0, 15.0, 7.5)):
        self._h = _lib3d().cmx_synth3d_world_create(seed, *size)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib3d().cmx_synth3d_world_destroy(self._h)
            self._h = None

    def free_position(self, seed, clearance=0.5):
        p = np.empty(3, np.float64)
        _lib3d().cmx_synth3d_free_position(self._h, seed, clearance, p)
        return p

    def scan(self, position, yaw, rings=16, azimuths=256, elev=np.deg2rad(15.0), max_range=30.0,
             sigma=0.01, seed=0):
        out = np.empty((rings * azimuths, 3), np.float32)
        n = _lib3d().cmx_synth3d_scan(self._h, np.ascontiguousarray(position, np.float64), yaw,
                                      rings, azimuths, elev, max_range, sigma, seed, out)
        return out[:n].copy()


def make_submap_3d(seed, resolution=0.1, size=(15.0, 15.0, 7.5), num_poses=8, rings=16,
                   azimuths=256, num_free_space_voxels=2):
    """Returns (HybridGrid, World3D): a box room rendered from `num_poses` scans."""
    world = World3D(seed, size)
    grid = HybridGrid(resolution)
    for p in range(num_poses):
        pos = world.free_position(seed * 1009 + p, 0.5)
        yaw = 0.37 * p
        sensor = world.scan(pos, yaw, rings, azimuths, seed=seed * 31 + p)
        c, s = np.cos(yaw), np.sin(yaw)
        in_map = sensor.astype(np.float64)
        x = pos[0] + c * in_map[:, 0] - s * in_map[:, 1]
        y = pos[1] + s * in_map[:, 0] + c * in_map[:, 1]
        z = pos[2] + in_map[:, 2]
        grid.insert(pos.astype(np.float32), np.stack([x, y, z], 1).astype(np.float32),
                    0.7, 0.4, num_free_space_voxels)
    return grid, world


def threaded_full_submap_searches(matchers, clouds, min_score, threads, calls_per_thread):
    """`threads` native threads (csrc/host/thread_driver.cc: a C++ caller's thread pool in
    miniature) each issue `calls_per_thread` cmx_fast2d_match_full_submap_batch_resident calls of
    `matchers` (scan_matching.FastCorrelativeScanMatcher2D) against `clouds` (PointCloudOnDevice,
    round-robin).  Returns (wall seconds inside the driver, candidates scored, matches found)."""
    from . import _lib as product
    entry = C.cast(product.lib().cmx_fast2d_match_full_submap_batch_resident, C.c_void_p)
    handles = (C.c_void_p * len(matchers))(*[m._h for m in matchers])
    cloud_handles = (C.c_void_p * len(clouds))(*[c._h for c in clouds])
    cand, found = C.c_int64(), C.c_int64()
    seconds = lib().cmx_thread_driver_fast2d(entry, handles, len(matchers), cloud_handles,
                                             len(clouds), min_score, threads, calls_per_thread,
                                             C.byref(cand), C.byref(found))
    if seconds < 0:
        raise RuntimeError(f"a search failed inside the thread driver: status {int(-seconds)}")
    return seconds, cand.value, found.value
