"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with oracle/Makefile (g++)."""
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


def _declare(L):
    L.orc_value_tables.argtypes = [_f32p, _f32p, _f32p]
    L.orc_probability_to_value.argtypes = [C.c_float]
    L.orc_correspondence_cost_to_value.argtypes = [C.c_float]
    L.orc_search_parameters.argtypes = [C.c_double, C.c_double, _f32p, C.c_int, C.c_double,
                                        C.POINTER(C.c_int), C.POINTER(C.c_double),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_generate_rotated_scans.argtypes = [_f32p, C.c_int, C.c_int, C.c_double, _f32p]
    L.orc_discretize_scans.argtypes = [_f32p, C.c_int, C.c_double, C.c_int, C.c_double,
                                       C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                       C.c_float, C.c_float, _i32p]
    L.orc_candidate2d.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                  C.c_int, _f64p]
    L.orc_grid_probability.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_grid_probability.restype = C.c_float
    L.orc_rt2d_match.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                 _f64p, _f32p, C.c_int, C.c_double, C.c_double, C.c_double,
                                 C.c_double, _f64p, C.POINTER(C.c_int64), C.c_void_p, C.c_int]
    L.orc_rt2d_match.restype = C.c_double
    L.orc_rt2d_match_tsdf.argtypes = [_u16p, _u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_double, C.c_float, C.c_float, _f64p, _f32p, C.c_int,
                                      C.c_double, C.c_double, C.c_double, C.c_double, _f64p,
                                      C.POINTER(C.c_int64), C.c_void_p, C.c_int]
    L.orc_rt2d_match_tsdf.restype = C.c_double
    L.orc_tsd_float_to_value.argtypes = [C.c_int, C.c_float, C.c_float]
    L.orc_tsd_value_to_float.argtypes = [C.c_int, C.c_float, C.c_int]
    L.orc_tsd_value_to_float.restype = C.c_float
    L.orc_fast2d_create.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                    C.c_double, C.c_int, C.c_double, C.c_double]
    L.orc_fast2d_create.restype = C.c_void_p
    L.orc_fast2d_destroy.argtypes = [C.c_void_p]
    L.orc_fast2d_level_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    L.orc_fast2d_level_cells.argtypes = [C.c_void_p, C.c_int, _u8p]
    L.orc_precompute2d.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, _u8p]
    L.orc_precompute2d_range.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         _u8p]
    L.orc_fast2d_match.argtypes = [C.c_void_p, _f64p, _f32p, C.c_int, C.c_int, C.c_float,
                                   C.POINTER(C.c_float), _f64p, _i64p]
    L.orc_fast2d_prepare.argtypes = [C.c_void_p, _f64p, _f32p, C.c_int, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.POINTER(C.c_int64)]
    L.orc_fast2d_match_batch.argtypes = [C.POINTER(C.c_void_p), C.c_int, _f32p, C.c_int,
                                         C.c_float, C.c_int, _i32p, _f32p, _f64p,
                                         C.POINTER(C.c_int64)]


# ---- oracle/_ref: reference translation units compiled in place (ref_shims/README.md) ------
_REF = os.path.join(_HERE, "_ref", "libref.so")
_ref_lib = None


def build_ref(reference="/root/reference"):
    """`make -C oracle ref` when the reference tree is present (it is not on the GPU box)."""
    if os.path.isdir(os.path.join(reference, "cartographer")):
        subprocess.check_call(["make", "-C", _HERE, "ref", f"REFERENCE={reference}"],
                              stdout=subprocess.DEVNULL)
    return os.path.exists(_REF)


_REF_CERES = os.path.join(_HERE, "_ref", "libref_ceres.so")
_ref_ceres_lib = None


def build_ref_ceres(reference="/root/reference"):
    """`make -C oracle ref_ceres` when the reference tree is present."""
    if os.path.isdir(os.path.join(reference, "cartographer")):
        subprocess.check_call(["make", "-C", _HERE, "ref_ceres", f"REFERENCE={reference}"],
                              stdout=subprocess.DEVNULL)
    return os.path.exists(_REF_CERES)


def ref_ceres_lib():
    """The reference's own Ceres-side sources (cost functions, CeresScanMatcher2D / 3D::Match)
    compiled in place over the stand-in solver of ref_shims/ceres/, or None."""
    global _ref_ceres_lib
    if _ref_ceres_lib is None and (os.path.exists(_REF_CERES) or build_ref_ceres()):
        L = C.CDLL(_REF_CERES)
        L.refc_ceres2d_match.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_double, _f64p, _f64p, _f64p, _f32p, C.c_int, _f64p,
                                         _f64p]
        L.refc_ceres2d_match.restype = None
        L.refc_ceres2d_residuals.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.c_double, _f64p, _f64p, C.c_double, _f64p, _f32p,
                                             C.c_int, _f64p, _f64p]
        L.refc_ceres2d_residuals.restype = None
        L.refc_ceres3d_match.argtypes = [_f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, _f64p, _f64p, _f64p, _f64p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refc_ceres3d_match.restype = None
        L.refc_ceres3d_residuals.argtypes = [_f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, _f64p, _f64p, _f64p, _f64p,
                                             _f64p]
        L.refc_ceres3d_residuals.restype = None
        L.refc_intensity3d_residuals.argtypes = [C.c_double, C.c_float, _f32p, _f32p, C.c_int,
                                                 C.c_float, C.c_void_p, C.c_int64, _f64p, _f64p,
                                                 _f64p]
        L.refc_intensity3d_residuals.restype = None
        _ref_ceres_lib = L
    return _ref_ceres_lib


def ref_lib():
    """The reference's own probability_values / value_conversion_tables / ray_to_pixel_mask
    code, or None when it has not been built (no /root/reference)."""
    global _ref_lib
    if _ref_lib is None and (os.path.exists(_REF) or build_ref()):
        L = C.CDLL(_REF)
        L.ref_ray_to_pixel_mask.argtypes = [C.c_int] * 5 + [_i32p, C.c_int]
        L.ref_value_tables.argtypes = [_f32p, _f32p]
        L.ref_probability_to_value.argtypes = [C.c_float]
        L.ref_correspondence_cost_to_value.argtypes = [C.c_float]
        L.ref_odds_tables.argtypes = [C.c_float, _u16p, _u16p]
        L.ref_conversion_table.argtypes = [C.c_float, C.c_float, C.c_float, _f32p]
        L.ref_tsd_float_to_value.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        L.ref_tsd_value_to_float.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int]
        L.ref_tsd_value_to_float.restype = C.c_float
        L.ref_fixed_ratio_sampler.argtypes = [C.c_double, C.c_int, _u8p]
        L.ref_fast2d_create.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_double, C.c_int, C.c_double, C.c_double]
        L.ref_fast2d_create.restype = C.c_void_p
        L.ref_fast2d_destroy.argtypes = [C.c_void_p]
        L.ref_fast2d_match.argtypes = [C.c_void_p, C.c_int, _f64p, _f32p, C.c_int, C.c_float,
                                       C.POINTER(C.c_float), _f64p]
        L.ref_precompute2d.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, _u8p]
        L.ref_precompute2d_tsdf.argtypes = [_u16p, _u16p, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_float, _u8p]
        L.ref_rt2d_match.argtypes = [_u16p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                     C.c_double, C.c_float, C.c_float, _f64p, _f32p, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _f64p]
        L.ref_rt2d_match.restype = C.c_double
        L.ref_grid2d_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                        C.c_void_p]
        L.ref_grid2d_create.restype = C.c_void_p
        L.ref_grid2d_destroy.argtypes = [C.c_void_p]
        L.ref_grid2d_get_limits.argtypes = [C.c_void_p, _f64p, _i32p]
        L.ref_grid2d_download.argtypes = [C.c_void_p, _u16p]
        L.ref_grid2d_insert.argtypes = [C.c_void_p, _f32p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_int, C.c_double, C.c_double, C.c_int]
        L.ref_grid2d_crop.argtypes = [C.c_void_p]
        L.ref_grid2d_set_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.ref_grid2d_get_probability.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_grid2d_get_probability.restype = C.c_float
        L.ref_map_limits_cell_index.argtypes = [C.c_double, C.c_double, C.c_double, _f32p,
                                                C.c_int, _i32p]
        L.ref_tsdf_create.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                      C.c_float, C.c_float]
        L.ref_tsdf_create.restype = C.c_void_p
        L.ref_tsdf_destroy.argtypes = [C.c_void_p]
        L.ref_tsdf_get_limits.argtypes = [C.c_void_p, _f64p, _i32p]
        L.ref_tsdf_download.argtypes = [C.c_void_p, _u16p, _u16p]
        L.ref_tsdf_insert.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, _f64p]
        # 3D: same layouts as the orc_*3d functions (_lib3d below).
        L.ref_grid3d_size.argtypes = [C.c_float, C.c_void_p, C.c_int64]
        L.ref_grid3d_iterate.argtypes = [C.c_float, C.c_void_p, C.c_int64, _i32p, C.c_int64]
        L.ref_grid3d_iterate.restype = C.c_int64
        L.ref_grid3d_cell_index.argtypes = [C.c_float, _f32p, C.c_int, _i32p]
        L.ref_hgrid_create.argtypes = [C.c_float]
        L.ref_hgrid_create.restype = C.c_void_p
        L.ref_hgrid_destroy.argtypes = [C.c_void_p]
        L.ref_hgrid_size.argtypes = [C.c_void_p]
        L.ref_hgrid_set_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.ref_hgrid_get_probability.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ref_hgrid_get_probability.restype = C.c_float
        L.ref_hgrid_insert.argtypes = [C.c_void_p, _f32p, _f32p, C.c_int, C.c_double, C.c_double,
                                       C.c_int]
        L.ref_hgrid_voxels.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_hgrid_voxels.restype = C.c_int64
        L.ref_igrid_create.argtypes = [C.c_float]
        L.ref_igrid_create.restype = C.c_void_p
        L.ref_igrid_destroy.argtypes = [C.c_void_p]
        L.ref_hgrid_insert_with_intensities.argtypes = [C.c_void_p, C.c_void_p, _f32p, _f32p,
                                                        C.c_void_p, C.c_int, C.c_double, C.c_double,
                                                        C.c_int, C.c_double]
        L.ref_igrid_voxels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_igrid_voxels.restype = C.c_int64
        L.ref_rt3d_match.argtypes = [C.c_float, C.c_void_p, C.c_int64, _f64p, _f32p, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _f64p,
                                     C.POINTER(C.c_int64)]
        L.ref_rt3d_match.restype = C.c_float
        L.ref_rt3d_match_mt.argtypes = [C.c_float, C.c_void_p, C.c_int64, _f64p, _f32p, C.c_int,
                                        C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                        _f64p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                        C.c_void_p, C.c_int64, C.c_int64]
        L.ref_rt3d_match_mt.restype = C.c_float
        L.ref_rotational_match.argtypes = [_f32p, _f32p, C.c_int, C.c_float, _f32p, C.c_int, _f32p]
        L.ref_compute_histogram.argtypes = [_f32p, C.c_int, C.c_int, _f32p]
        L.ref_fast3d_create.argtypes = [C.c_float, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                        C.c_int64, _f32p, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_double, C.c_double, C.c_double]
        L.ref_fast3d_create.restype = C.c_void_p
        L.ref_fast3d_destroy.argtypes = [C.c_void_p]
        L.ref_fast3d_level_count.argtypes = [C.c_void_p, C.c_int]
        L.ref_fast3d_level_count.restype = C.c_int64
        L.ref_fast3d_level_voxels.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.ref_fast3d_match.argtypes = [C.c_void_p, C.c_int, _f64p, _f64p, _f64p, _f32p, C.c_int,
                                       _f32p, C.c_int, _f32p, C.c_int, C.c_float, _f64p, _i64p]
        _ref_lib = L
    return _ref_lib


class ReferenceProbabilityGrid:
    """The reference's own ProbabilityGrid (grid_2d.cc, probability_grid.cc) driven by its own
    ProbabilityGridRangeDataInserter2D; same surface as cartographer_amd.synth.ProbabilityGrid."""

    def __init__(self, resolution, max_xy, num_x_cells, num_y_cells, cells=None):
        c = None
        if cells is not None:
            self._cells_in = np.ascontiguousarray(cells, np.uint16)
            c = self._cells_in.ctypes.data
        self._h = ref_lib().ref_grid2d_create(resolution, max_xy[0], max_xy[1], num_x_cells,
                                              num_y_cells, c)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_grid2d_destroy(self._h)
            self._h = None

    @property
    def limits(self):
        lim = np.empty(3, np.float64)
        n = np.empty(2, np.int32)
        ref_lib().ref_grid2d_get_limits(self._h, lim, n)
        return dict(resolution=float(lim[0]), max_x=float(lim[1]), max_y=float(lim[2]),
                    num_x_cells=int(n[0]), num_y_cells=int(n[1]))

    @property
    def cells(self):
        lim = self.limits
        out = np.empty((lim["num_y_cells"], lim["num_x_cells"]), np.uint16)
        ref_lib().ref_grid2d_download(self._h, out)
        return out

    def insert(self, origin_xy, returns_xyz, misses_xyz=None, hit_probability=0.7,
               miss_probability=0.4, insert_free_space=True):
        origin = np.ascontiguousarray(origin_xy, np.float32)[:2].copy()
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        mis = (np.ascontiguousarray(misses_xyz, np.float32).reshape(-1, 3)
               if misses_xyz is not None else np.zeros((0, 3), np.float32))
        ref_lib().ref_grid2d_insert(self._h, origin, ret.ctypes.data, ret.shape[0],
                                    mis.ctypes.data, mis.shape[0], hit_probability,
                                    miss_probability, int(insert_free_space))

    def set_probability(self, ix, iy, probability):
        ref_lib().ref_grid2d_set_probability(self._h, ix, iy, probability)

    def get_probability(self, ix, iy):
        return float(ref_lib().ref_grid2d_get_probability(self._h, ix, iy))

    def crop(self):
        ref_lib().ref_grid2d_crop(self._h)


class ReferenceTSDF2D:
    """The reference's own TSDF2D (tsdf_2d.cc) filled by its own TSDFRangeDataInserter2D
    (tsdf_range_data_inserter_2d.cc + normal_estimation_2d.cc)."""

    def __init__(self, resolution, max_xy, num_x_cells, num_y_cells, truncation_distance,
                 max_weight):
        self.truncation_distance, self.max_weight = truncation_distance, max_weight
        self._h = ref_lib().ref_tsdf_create(resolution, max_xy[0], max_xy[1], num_x_cells,
                                            num_y_cells, truncation_distance, max_weight)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_tsdf_destroy(self._h)
            self._h = None

    @property
    def limits(self):
        lim = np.empty(3, np.float64)
        n = np.empty(2, np.int32)
        ref_lib().ref_tsdf_get_limits(self._h, lim, n)
        return dict(resolution=float(lim[0]), max_x=float(lim[1]), max_y=float(lim[2]),
                    num_x_cells=int(n[0]), num_y_cells=int(n[1]))

    def planes(self):
        """(tsd cells, weight cells), uint16 [ny, nx] each."""
        lim = self.limits
        shape = (lim["num_y_cells"], lim["num_x_cells"])
        tsd, wgt = np.empty(shape, np.uint16), np.empty(shape, np.uint16)
        ref_lib().ref_tsdf_download(self._h, tsd, wgt)
        return tsd, wgt

    def insert(self, origin_xyz, returns_xyz, truncation_distance, maximum_weight,
               update_free_space, num_normal_samples, sample_radius,
               project_sdf_distance_to_scan_normal, update_weight_range_exponent,
               angle_kernel_bandwidth, distance_kernel_bandwidth):
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        options = np.array([truncation_distance, maximum_weight, float(update_free_space),
                            num_normal_samples, sample_radius,
                            float(project_sdf_distance_to_scan_normal),
                            update_weight_range_exponent, angle_kernel_bandwidth,
                            distance_kernel_bandwidth], np.float64)
        ref_lib().ref_tsdf_insert(self._h, np.ascontiguousarray(origin_xyz, np.float32), ret,
                                  ret.shape[0], options)


def ref_map_limits_cell_index(resolution, max_x, max_y, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.empty((xy.shape[0], 2), np.int32)
    ref_lib().ref_map_limits_cell_index(resolution, max_x, max_y, xy, xy.shape[0], out)
    return out


class ReferenceFastCorrelativeScanMatcher2D:
    """The reference's own fast_correlative_scan_matcher_2d.cc / correlative_scan_matcher_2d.cc
    (oracle/_ref), same call shape as FastCorrelativeScanMatcher2D below."""

    def __init__(self, cells, res, max_x, max_y, depth, linear_search_window=7.0,
                 angular_search_window=float(np.deg2rad(30.0))):
        cells = np.ascontiguousarray(cells, np.uint16)
        ny, nx = cells.shape
        self._h = ref_lib().ref_fast2d_create(cells, nx, ny, res, max_x, max_y, depth,
                                              linear_search_window, angular_search_window)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_fast2d_destroy(self._h)
            self._h = None

    def _match(self, init, xyz, full, min_score):
        xyz, n = _cloud(xyz)
        score = C.c_float()
        pose = np.zeros(3, np.float64)
        ok = ref_lib().ref_fast2d_match(self._h, int(full), np.ascontiguousarray(init, np.float64),
                                        xyz, n, min_score, C.byref(score), pose)
        return dict(found=bool(ok), score=float(score.value), pose=pose)

    def match(self, init_xyt, xyz, min_score):
        return self._match(init_xyt, xyz, False, min_score)

    def match_full_submap(self, xyz, min_score):
        return self._match([0.0, 0.0, 0.0], xyz, True, min_score)


def ref_precompute2d(cells, width):
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    out = np.empty((ny + width - 1, nx + width - 1), np.uint8)
    ref_lib().ref_precompute2d(cells, nx, ny, width, out)
    return out


def ref_precompute2d_tsdf(tsd_cells, weight_cells, width, truncation_distance, max_weight):
    """The reference's PrecomputationGrid2D over its own TSDF2D (two uint16 planes)."""
    tsd = np.ascontiguousarray(tsd_cells, np.uint16)
    wgt = np.ascontiguousarray(weight_cells, np.uint16)
    ny, nx = tsd.shape
    out = np.empty((ny + width - 1, nx + width - 1), np.uint8)
    ref_lib().ref_precompute2d_tsdf(tsd, wgt, nx, ny, width, truncation_distance, max_weight, out)
    return out


def ref_rt2d_match(cells, res, max_x, max_y, init_xyt, xyz, lin, ang, tw, rw, weight_cells=None,
                   truncation_distance=0.0, max_weight=0.0):
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    wptr = None
    if weight_cells is not None:
        weight_cells = np.ascontiguousarray(weight_cells, np.uint16)
        wptr = weight_cells.ctypes.data
    xyz, n = _cloud(xyz)
    pose = np.zeros(3, np.float64)
    s = ref_lib().ref_rt2d_match(cells, wptr, nx, ny, res, max_x, max_y, truncation_distance,
                                 max_weight, np.ascontiguousarray(init_xyt, np.float64), xyz, n,
                                 lin, ang, tw, rw, pose)
    return dict(score=float(s), pose=pose)


def _cloud(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    return xyz, xyz.shape[0]


def value_tables():
    a = np.empty(65536, np.float32)
    b = np.empty(65536, np.float32)
    c = np.empty(65536, np.float32)
    lib().orc_value_tables(a, b, c)
    return a, b, c


def search_parameters(lin, ang, xyz, res):
    xyz, n = _cloud(xyz)
    na, step, ns, nl = C.c_int(), C.c_double(), C.c_int(), C.c_int()
    lib().orc_search_parameters(lin, ang, xyz, n, res, C.byref(na), C.byref(step),
                                C.byref(ns), C.byref(nl))
    return dict(num_angular_perturbations=na.value, angular_perturbation_step_size=step.value,
                num_scans=ns.value, num_linear_perturbations=nl.value)


def generate_rotated_scans(xyz, na, step):
    xyz, n = _cloud(xyz)
    out = np.empty((2 * na + 1, n, 3), np.float32)
    lib().orc_generate_rotated_scans(xyz, n, na, step, out)
    return out


def discretize_scans(xyz, init_theta, na, step, res, max_x, max_y, nx, ny, tx=0.0, ty=0.0):
    xyz, n = _cloud(xyz)
    out = np.empty((2 * na + 1, n, 2), np.int32)
    lib().orc_discretize_scans(xyz, n, init_theta, na, step, res, max_x, max_y, nx, ny,
                               tx, ty, out)
    return out


def candidate2d(nl, na, step, res, scan_index, x_off, y_off):
    out = np.empty(3, np.float64)
    lib().orc_candidate2d(nl, na, step, res, scan_index, x_off, y_off, out)
    return out


def grid_probability(cells, ix, iy):
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    return float(lib().orc_grid_probability(cells, nx, ny, ix, iy))


def rt2d_match(cells, res, max_x, max_y, init_xyt, xyz, lin, ang, tw, rw, want_scores=False):
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    xyz, n = _cloud(xyz)
    pose = np.empty(3, np.float64)
    ncand = C.c_int64()
    init = np.ascontiguousarray(init_xyt, np.float64)
    scores = None
    if want_scores:
        # first call to learn the count
        s = lib().orc_rt2d_match(cells, nx, ny, res, max_x, max_y, init, xyz, n, lin, ang, tw,
                                 rw, pose, C.byref(ncand), None, 0)
        scores = np.empty(ncand.value, np.float32)
        s = lib().orc_rt2d_match(cells, nx, ny, res, max_x, max_y, init, xyz, n, lin, ang, tw,
                                 rw, pose, C.byref(ncand), scores.ctypes.data, scores.size)
    else:
        s = lib().orc_rt2d_match(cells, nx, ny, res, max_x, max_y, init, xyz, n, lin, ang, tw,
                                 rw, pose, C.byref(ncand), None, 0)
    return dict(score=float(s), pose=pose, num_candidates=ncand.value, scores=scores)


def _ceres_options(occupied_space_weight, translation_weight, rotation_weight,
                   use_nonmonotonic_steps, max_num_iterations):
    return np.array([occupied_space_weight, translation_weight, rotation_weight,
                     1.0 if use_nonmonotonic_steps else 0.0, max_num_iterations], np.float64)


def ceres2d_match(cells, res, max_x, max_y, target_xy, init_xyt, xyz, occupied_space_weight=1.0,
                  translation_weight=10.0, rotation_weight=40.0, use_nonmonotonic_steps=False,
                  max_num_iterations=20, reference=False):
    """CeresScanMatcher2D::Match restated (oracle_ceres_2d.h: Ceres itself is absent, its
    published trust-region algorithm is restated; parity with Ceres's iterates is unpinned)."""
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    xyz, n = _cloud(xyz)
    L = lib()
    L.orc_ceres2d_match.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                    _f64p, _f64p, _f64p, _f32p, C.c_int, _f64p, _f64p]
    L.orc_ceres2d_match.restype = None
    pose, summary = np.empty(3, np.float64), np.empty(5, np.float64)
    fn = ref_ceres_lib().refc_ceres2d_match if reference else L.orc_ceres2d_match
    fn(cells, nx, ny, res, max_x, max_y,
                        _ceres_options(occupied_space_weight, translation_weight, rotation_weight,
                                       use_nonmonotonic_steps, max_num_iterations),
                        np.ascontiguousarray(target_xy, np.float64),
                        np.ascontiguousarray(init_xyt, np.float64), xyz, n, pose, summary)
    return dict(pose=pose, initial_cost=summary[0], final_cost=summary[1],
                num_successful_steps=int(summary[2]), num_unsuccessful_steps=int(summary[3]),
                termination=int(summary[4]))


def ceres2d_residuals(cells, res, max_x, max_y, target_xy, target_angle, pose_xyt, xyz,
                      occupied_space_weight=1.0, translation_weight=10.0, rotation_weight=40.0,
                      reference=False):
    """Residuals [n + 3] and Jacobian [n + 3, 3] of the three residual blocks at `pose_xyt`."""
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    xyz, n = _cloud(xyz)
    L = lib()
    L.orc_ceres2d_residuals.argtypes = [_u16p, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_double, _f64p, _f64p, C.c_double, _f64p, _f32p,
                                        C.c_int, _f64p, _f64p]
    L.orc_ceres2d_residuals.restype = None
    r, J = np.empty(n + 3, np.float64), np.empty((n + 3, 3), np.float64)
    fn = ref_ceres_lib().refc_ceres2d_residuals if reference else L.orc_ceres2d_residuals
    fn(cells, nx, ny, res, max_x, max_y,
                            _ceres_options(occupied_space_weight, translation_weight,
                                           rotation_weight, False, 0),
                            np.ascontiguousarray(target_xy, np.float64), target_angle,
                            np.ascontiguousarray(pose_xyt, np.float64), xyz, n, r, J)
    return r, J


def rt2d_match_tsdf(tsd_cells, weight_cells, res, max_x, max_y, truncation_distance, max_weight,
                    init_xyt, xyz, lin, ang, tw, rw, want_scores=False):
    """RealTimeCorrelativeScanMatcher2D::Match on a TSDF2D (two uint16 planes)."""
    tsd = np.ascontiguousarray(tsd_cells, np.uint16)
    wgt = np.ascontiguousarray(weight_cells, np.uint16)
    ny, nx = tsd.shape
    assert wgt.shape == tsd.shape
    xyz, n = _cloud(xyz)
    pose = np.empty(3, np.float64)
    ncand = C.c_int64()
    init = np.ascontiguousarray(init_xyt, np.float64)
    args = (tsd, wgt, nx, ny, res, max_x, max_y, truncation_distance, max_weight, init, xyz, n,
            lin, ang, tw, rw, pose, C.byref(ncand))
    s = lib().orc_rt2d_match_tsdf(*args, None, 0)
    scores = None
    if want_scores:
        scores = np.empty(ncand.value, np.float32)
        s = lib().orc_rt2d_match_tsdf(*args, scores.ctypes.data, scores.size)
    return dict(score=float(s), pose=pose, num_candidates=ncand.value, scores=scores)


def tsd_to_value(tsd, truncation_distance):
    return lib().orc_tsd_float_to_value(0, truncation_distance, tsd)


def weight_to_value(weight, max_weight):
    return lib().orc_tsd_float_to_value(1, max_weight, weight)


def value_to_tsd(value, truncation_distance):
    return float(lib().orc_tsd_value_to_float(0, truncation_distance, value))


def value_to_weight(value, max_weight):
    return float(lib().orc_tsd_value_to_float(1, max_weight, value))


def precompute2d(cells, width):
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    out = np.empty((ny + width - 1, nx + width - 1), np.uint8)
    lib().orc_precompute2d(cells, nx, ny, width, out)
    return out


def precompute2d_range(cells, width, min_correspondence_cost, max_correspondence_cost):
    """PrecomputationGrid2D over a Grid2D with other correspondence-cost bounds (a TSDF2D's tsd
    plane: -truncation_distance, truncation_distance)."""
    cells = np.ascontiguousarray(cells, np.uint16)
    ny, nx = cells.shape
    out = np.empty((ny + width - 1, nx + width - 1), np.uint8)
    lib().orc_precompute2d_range(cells, nx, ny, width, min_correspondence_cost,
                                 max_correspondence_cost, out)
    return out


class FastCorrelativeScanMatcher2D:
    """Oracle twin of fast_correlative_scan_matcher_2d.h:114-136."""

    def __init__(self, cells, res, max_x, max_y, depth, linear_search_window=7.0,
                 angular_search_window=np.deg2rad(30.0)):
        cells = np.ascontiguousarray(cells, np.uint16)
        self.ny, self.nx = cells.shape
        self.depth = depth
        self._h = lib().orc_fast2d_create(cells, self.nx, self.ny, res, max_x, max_y, depth,
                                          linear_search_window, angular_search_window)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fast2d_destroy(self._h)
            self._h = None

    def level(self, i):
        wx, wy = C.c_int(), C.c_int()
        lib().orc_fast2d_level_dims(self._h, i, C.byref(wx), C.byref(wy))
        out = np.empty((wy.value, wx.value), np.uint8)
        lib().orc_fast2d_level_cells(self._h, i, out)
        return out

    def _match(self, init, xyz, full, min_score):
        xyz, n = _cloud(xyz)
        score = C.c_float()
        pose = np.zeros(3, np.float64)
        stats = np.zeros(4, np.int64)
        ok = lib().orc_fast2d_match(self._h, np.ascontiguousarray(init, np.float64), xyz, n,
                                    int(full), min_score, C.byref(score), pose, stats)
        return dict(found=bool(ok), score=float(score.value), pose=pose,
                    candidates_scored=int(stats[0]), num_scans=int(stats[1]),
                    coarse_candidates=int(stats[2]), nodes_expanded=int(stats[3]))

    def match(self, init_xyt, xyz, min_score):
        return self._match(init_xyt, xyz, False, min_score)

    def match_full_submap(self, xyz, min_score):
        return self._match([0.0, 0.0, 0.0], xyz, True, min_score)

    def prepare(self, init_xyt, xyz, full, want_sums=True):
        xyz, n = _cloud(xyz)
        init = np.ascontiguousarray(init_xyt, np.float64)
        ns, step, nsums = C.c_int(), C.c_double(), C.c_int64()
        rc = lib().orc_fast2d_prepare(self._h, init, xyz, n, int(full), C.byref(ns),
                                      C.byref(step), None, 0, None, 0, None, 0,
                                      C.byref(nsums) if want_sums else None)
        assert rc == 0
        scans = np.empty((ns.value, n, 2), np.int32)
        bounds = np.empty((ns.value, 4), np.int32)
        sums = np.empty(nsums.value if want_sums else 0, np.int32)
        rc = lib().orc_fast2d_prepare(self._h, init, xyz, n, int(full), C.byref(ns),
                                      C.byref(step), scans.ctypes.data, scans.size,
                                      bounds.ctypes.data, bounds.size,
                                      sums.ctypes.data if want_sums else None, sums.size,
                                      C.byref(nsums) if want_sums else None)
        assert rc == 0
        return dict(num_scans=ns.value, step=step.value, scans=scans, bounds=bounds, sums=sums)


def fast2d_match_batch(matchers, xyz, min_score, num_threads):
    xyz, n = _cloud(xyz)
    num = len(matchers)
    handles = (C.c_void_p * num)(*[m._h for m in matchers])
    found = np.zeros(num, np.int32)
    scores = np.zeros(num, np.float32)
    poses = np.zeros((num, 3), np.float64)
    total = C.c_int64()
    lib().orc_fast2d_match_batch(handles, num, xyz, n, min_score, num_threads, found, scores,
                                 poses, C.byref(total))
    return dict(found=found, scores=scores, poses=poses, candidates_scored=total.value)


# ---------------------------------------------------------------- 3D -------
VOXEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32), ("value", np.uint16),
                        ("pad", np.uint16)])


def _lib3d():
    L = lib()
    if not getattr(L, "_orc_3d_declared", False):
        L.orc_grid3d_size.argtypes = [C.c_float, C.c_void_p, C.c_int64]
        L.orc_rt3d_match.argtypes = [C.c_float, C.c_void_p, C.c_int64, _f64p, _f32p, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _f64p,
                                     C.POINTER(C.c_int64)]
        L.orc_rt3d_match.restype = C.c_float
        L.orc_rt3d_match_mt.argtypes = [C.c_float, C.c_void_p, C.c_int64, _f64p, _f32p, C.c_int,
                                        C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                        _f64p, C.POINTER(C.c_int64)]
        L.orc_rt3d_match_mt.restype = C.c_float
        L.orc_rotational_match.argtypes = [_f32p, _f32p, C.c_int, C.c_float, _f32p, C.c_int, _f32p]
        L.orc_fast3d_create.argtypes = [C.c_float, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                        C.c_int64, _f32p, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_fast3d_create.restype = C.c_void_p
        L.orc_fast3d_destroy.argtypes = [C.c_void_p]
        L.orc_fast3d_level_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_fast3d_level_count.restype = C.c_int64
        L.orc_fast3d_level_voxels.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.orc_fast3d_match.argtypes = [C.c_void_p, C.c_int, _f64p, _f64p, _f64p, _f32p, C.c_int,
                                       _f32p, C.c_int, _f32p, C.c_int, C.c_float, _f64p, _i64p]
        L._orc_3d_declared = True
    return L


def _voxels(v):
    v = np.ascontiguousarray(v, VOXEL_DTYPE)
    return v, v.shape[0]


def grid3d_size(resolution, voxels):
    v, n = _voxels(voxels)
    return int(_lib3d().orc_grid3d_size(resolution, v.ctypes.data, n))


def rt3d_match(resolution, voxels, init_pose7, xyz, lin, ang, tw, rw, num_threads=1):
    """init_pose7 = (tx,ty,tz, qw,qx,qy,qz).  num_threads > 1 spreads the z slices of the window
    over host threads (same result: joined in z order with the first-maximum rule)."""
    v, n = _voxels(voxels)
    xyz, npts = _cloud(xyz)
    pose = np.empty(7, np.float64)
    ncand = C.c_int64()
    if num_threads > 1:
        s = _lib3d().orc_rt3d_match_mt(resolution, v.ctypes.data, n,
                                       np.ascontiguousarray(init_pose7, np.float64), xyz, npts,
                                       lin, ang, tw, rw, int(num_threads), pose, C.byref(ncand))
        return dict(score=float(s), pose=pose, num_candidates=ncand.value)
    s = _lib3d().orc_rt3d_match(resolution, v.ctypes.data, n,
                                np.ascontiguousarray(init_pose7, np.float64), xyz, npts, lin, ang,
                                tw, rw, pose, C.byref(ncand))
    return dict(score=float(s), pose=pose, num_candidates=ncand.value)


def rotational_match(submap_hist, scan_hist, initial_angle, angles):
    a = np.ascontiguousarray(submap_hist, np.float32)
    b = np.ascontiguousarray(scan_hist, np.float32)
    ang = np.ascontiguousarray(angles, np.float32)
    out = np.empty(ang.shape[0], np.float32)
    _lib3d().orc_rotational_match(a, b, a.shape[0], initial_angle, ang, ang.shape[0], out)
    return out


class FastCorrelativeScanMatcher3D:
    """Oracle twin of fast_correlative_scan_matcher_3d.h:75-101."""

    def __init__(self, resolution, voxels, low_resolution, low_voxels, histogram, depth,
                 full_resolution_depth, min_rotational_score, min_low_resolution_score,
                 linear_xy_search_window, linear_z_search_window, angular_search_window):
        v, n = _voxels(voxels)
        lv, nl = _voxels(low_voxels)
        h = np.ascontiguousarray(histogram, np.float32)
        self.depth = depth
        self._h = _lib3d().orc_fast3d_create(resolution, v.ctypes.data, n, low_resolution,
                                             lv.ctypes.data, nl, h, h.shape[0], depth,
                                             full_resolution_depth, min_rotational_score,
                                             min_low_resolution_score, linear_xy_search_window,
                                             linear_z_search_window, angular_search_window)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib3d().orc_fast3d_destroy(self._h)
            self._h = None

    def level(self, depth):
        """Non-zero cells of one precomputation level as an int32 [n,4] (x,y,z,value) array,
        sorted (z,y,x)."""
        n = _lib3d().orc_fast3d_level_count(self._h, depth)
        out = np.empty((n, 4), np.int32)
        if n:
            _lib3d().orc_fast3d_level_voxels(self._h, depth, out)
        return out

    def _match(self, full, node7, submap7, gravity, hi, lo, hist, min_score):
        hi, nhi = _cloud(hi)
        lo, nlo = _cloud(lo)
        hist = np.ascontiguousarray(hist, np.float32)
        res = np.zeros(10, np.float64)
        stats = np.zeros(4, np.int64)
        ok = _lib3d().orc_fast3d_match(self._h, int(full),
                                       np.ascontiguousarray(node7, np.float64),
                                       np.ascontiguousarray(submap7, np.float64),
                                       np.ascontiguousarray(gravity, np.float64), hi, nhi, lo,
                                       nlo, hist, hist.shape[0], min_score, res, stats)
        return dict(found=bool(ok), score=float(np.float32(res[0])), pose=res[1:8].copy(),
                    rotational_score=float(np.float32(res[8])),
                    low_resolution_score=float(np.float32(res[9])),
                    candidates_scored=int(stats[0]), num_scans=int(stats[1]),
                    coarse_candidates=int(stats[2]), nodes_expanded=int(stats[3]))

    def match(self, node7, submap7, gravity, hi, lo, hist, min_score):
        return self._match(False, node7, submap7, gravity, hi, lo, hist, min_score)

    def match_full_submap(self, node_q, submap_q, gravity, hi, lo, hist, min_score):
        node7 = np.concatenate([[0, 0, 0], node_q])
        submap7 = np.concatenate([[0, 0, 0], submap_q])
        return self._match(True, node7, submap7, gravity, hi, lo, hist, min_score)


# ---- the reference's own 3D sources (oracle/_ref), same call shapes as the oracle twins above ----
def voxel_filter_flags(xyz, resolution):
    """points_used flags of sensor::VoxelFilter (the randomised reservoir filter)."""
    xyz, n = _cloud(xyz)
    used = np.zeros(n, np.uint8)
    L = lib()
    L.orc_voxel_filter_flags.argtypes = [_f32p, C.c_int, C.c_float, C.c_void_p]
    L.orc_voxel_filter_flags.restype = None
    L.orc_voxel_filter_flags(xyz, n, resolution, used.ctypes.data)
    return used.astype(bool)


def adaptive_voxel_filter(xyz, max_length, min_num_points, max_range):
    xyz, n = _cloud(xyz)
    out = np.empty((max(n, 1), 3), np.float32)
    L = lib()
    L.orc_adaptive_voxel_filter.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p]
    L.orc_adaptive_voxel_filter.restype = C.c_int
    m = L.orc_adaptive_voxel_filter(xyz, n, max_length, min_num_points, max_range,
                                    out.ctypes.data)
    return out[:m].copy()


def compute_histogram(xyz, histogram_size):
    xyz, n = _cloud(xyz)
    out = np.empty(histogram_size, np.float32)
    L = lib()
    L.orc_compute_histogram.argtypes = [_f32p, C.c_int, C.c_int, C.c_void_p]
    L.orc_compute_histogram.restype = None
    L.orc_compute_histogram(xyz, n, histogram_size, out.ctypes.data)
    return out


def ref_voxel_filter(xyz, resolution):
    """The reference's own sensor/internal/voxel_filter.cc (oracle/_ref)."""
    xyz, n = _cloud(xyz)
    out = np.empty((max(n, 1), 3), np.float32)
    R = ref_lib()
    R.ref_voxel_filter.argtypes = [_f32p, C.c_int, C.c_float, C.c_void_p]
    R.ref_voxel_filter.restype = C.c_int
    return out[:R.ref_voxel_filter(xyz, n, resolution, out.ctypes.data)].copy()


def ref_adaptive_voxel_filter(xyz, max_length, min_num_points, max_range):
    xyz, n = _cloud(xyz)
    out = np.empty((max(n, 1), 3), np.float32)
    R = ref_lib()
    R.ref_adaptive_voxel_filter.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p]
    R.ref_adaptive_voxel_filter.restype = C.c_int
    m = R.ref_adaptive_voxel_filter(xyz, n, max_length, min_num_points, max_range,
                                    out.ctypes.data)
    return out[:m].copy()


def _sort_zyx(a):
    return a[np.lexsort((a[:, 0], a[:, 1], a[:, 2]))] if len(a) else a


def ref_grid3d_size(resolution, voxels):
    v, n = _voxels(voxels)
    return int(ref_lib().ref_grid3d_size(resolution, v.ctypes.data, n))


def ref_grid3d_iterate(resolution, voxels):
    """(x, y, z, value) rows the real HybridGrid's iterator yields after the voxels were written
    through mutable_value, in iteration order."""
    v, n = _voxels(voxels)
    out = np.empty((max(n, 1), 4), np.int32)
    k = ref_lib().ref_grid3d_iterate(resolution, v.ctypes.data, n, out, out.shape[0])
    return out[:k]


def ref_grid3d_cell_index(resolution, xyz):
    xyz, n = _cloud(xyz)
    out = np.empty((n, 3), np.int32)
    ref_lib().ref_grid3d_cell_index(resolution, xyz, n, out)
    return out


class ReferenceHybridGrid:
    """The reference's own HybridGrid (mapping/3d/hybrid_grid.h) driven by its own
    RangeDataInserter3D; same surface as cartographer_amd.synth.HybridGrid."""

    def __init__(self, resolution):
        self._h = ref_lib().ref_hgrid_create(resolution)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_hgrid_destroy(self._h)
            self._h = None

    @property
    def grid_size(self):
        return int(ref_lib().ref_hgrid_size(self._h))

    def set_probability(self, index, probability):
        ref_lib().ref_hgrid_set_probability(self._h, int(index[0]), int(index[1]),
                                            int(index[2]), probability)

    def get_probability(self, index):
        return float(ref_lib().ref_hgrid_get_probability(self._h, int(index[0]), int(index[1]),
                                                         int(index[2])))

    def insert(self, origin_xyz, returns_xyz, hit_probability=0.7, miss_probability=0.4,
               num_free_space_voxels=5):
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        ref_lib().ref_hgrid_insert(self._h, np.ascontiguousarray(origin_xyz, np.float32), ret,
                                   ret.shape[0], hit_probability, miss_probability,
                                   num_free_space_voxels)

    def insert_with_intensities(self, intensity_grid, origin_xyz, returns_xyz, intensities,
                                hit_probability=0.7, miss_probability=0.4, num_free_space_voxels=5,
                                intensity_threshold=40.0):
        """RangeDataInserter3D::Insert with an IntensityHybridGrid (the reference's own source)."""
        ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
        ints = None if intensities is None else np.ascontiguousarray(intensities, np.float32)
        ref_lib().ref_hgrid_insert_with_intensities(
            self._h, intensity_grid._h, np.ascontiguousarray(origin_xyz, np.float32), ret,
            None if ints is None else ints.ctypes.data, ret.shape[0], hit_probability,
            miss_probability, num_free_space_voxels, intensity_threshold)

    def voxels(self):
        """Non-zero cells as VOXEL_DTYPE records sorted (z, y, x), like synth.HybridGrid."""
        n = ref_lib().ref_hgrid_voxels(self._h, None, 0)
        rows = np.empty((max(n, 1), 4), np.int32)
        ref_lib().ref_hgrid_voxels(self._h, rows.ctypes.data, rows.shape[0])
        rows = _sort_zyx(rows[:n])
        out = np.zeros(n, VOXEL_DTYPE)
        out["x"], out["y"], out["z"], out["value"] = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
        return out


def ref_rt3d_match(resolution, voxels, init_pose7, xyz, lin, ang, tw, rw):
    v, n = _voxels(voxels)
    xyz, npts = _cloud(xyz)
    pose = np.empty(7, np.float64)
    ncand = C.c_int64()
    s = ref_lib().ref_rt3d_match(resolution, v.ctypes.data, n,
                                 np.ascontiguousarray(init_pose7, np.float64), xyz, npts, lin,
                                 ang, tw, rw, pose, C.byref(ncand))
    return dict(score=float(s), pose=pose)


def ref_rt3d_match_mt(resolution, voxels, init_pose7, xyz, lin, ang, tw, rw, num_threads=8,
                      first_candidate=0, max_candidates=0):
    """The reference's GenerateExhaustiveSearchTransforms / TransformPointCloud / ScoreCandidate
    over candidate ranges on `num_threads` host threads (oracle/ref_wrapper_rt3d_mt.cc).
    max_candidates > 0: only that many candidates from `first_candidate` on (a bounded sample)."""
    v, n = _voxels(voxels)
    xyz, npts = _cloud(xyz)
    pose = np.empty(7, np.float64)
    ncand, best = C.c_int64(), C.c_int64()
    s = ref_lib().ref_rt3d_match_mt(resolution, v.ctypes.data, n,
                                    np.ascontiguousarray(init_pose7, np.float64), xyz, npts, lin,
                                    ang, tw, rw, int(num_threads), pose, C.byref(best),
                                    C.byref(ncand), None, int(first_candidate),
                                    int(max_candidates))
    return dict(score=float(s), pose=pose, best_index=int(best.value),
                num_candidates=int(ncand.value))


def ref_rotational_match(submap_hist, scan_hist, initial_angle, angles):
    a = np.ascontiguousarray(submap_hist, np.float32)
    b = np.ascontiguousarray(scan_hist, np.float32)
    ang = np.ascontiguousarray(angles, np.float32)
    out = np.empty(ang.shape[0], np.float32)
    ref_lib().ref_rotational_match(a, b, a.shape[0], initial_angle, ang, ang.shape[0], out)
    return out


def ref_compute_histogram(xyz, histogram_size):
    xyz, n = _cloud(xyz)
    out = np.empty(histogram_size, np.float32)
    ref_lib().ref_compute_histogram(xyz, n, histogram_size, out)
    return out


class ReferenceFastCorrelativeScanMatcher3D:
    """The reference's own fast_correlative_scan_matcher_3d.cc (+ precomputation_grid_3d.cc,
    rotational_scan_matcher.cc, low_resolution_matcher.cc, the real hybrid_grid.h)."""

    def __init__(self, resolution, voxels, low_resolution, low_voxels, histogram, depth,
                 full_resolution_depth, min_rotational_score, min_low_resolution_score,
                 linear_xy_search_window, linear_z_search_window, angular_search_window):
        v, n = _voxels(voxels)
        lv, nl = _voxels(low_voxels)
        h = np.ascontiguousarray(histogram, np.float32)
        self.depth = depth
        self._h = ref_lib().ref_fast3d_create(resolution, v.ctypes.data, n, low_resolution,
                                              lv.ctypes.data, nl, h, h.shape[0], depth,
                                              full_resolution_depth, min_rotational_score,
                                              min_low_resolution_score, linear_xy_search_window,
                                              linear_z_search_window, angular_search_window)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_fast3d_destroy(self._h)
            self._h = None

    def level(self, depth):
        """Like FastCorrelativeScanMatcher3D.level: int32 [n,4], sorted (z,y,x)."""
        n = ref_lib().ref_fast3d_level_count(self._h, depth)
        out = np.empty((n, 4), np.int32)
        if n:
            ref_lib().ref_fast3d_level_voxels(self._h, depth, out)
        return _sort_zyx(out)

    def _match(self, full, node7, submap7, gravity, hi, lo, hist, min_score):
        hi, nhi = _cloud(hi)
        lo, nlo = _cloud(lo)
        hist = np.ascontiguousarray(hist, np.float32)
        res = np.zeros(10, np.float64)
        stats = np.zeros(4, np.int64)
        ok = ref_lib().ref_fast3d_match(self._h, int(full),
                                        np.ascontiguousarray(node7, np.float64),
                                        np.ascontiguousarray(submap7, np.float64),
                                        np.ascontiguousarray(gravity, np.float64), hi, nhi, lo,
                                        nlo, hist, hist.shape[0], min_score, res, stats)
        return dict(found=bool(ok), score=float(np.float32(res[0])), pose=res[1:8].copy(),
                    rotational_score=float(np.float32(res[8])),
                    low_resolution_score=float(np.float32(res[9])))

    def match(self, node7, submap7, gravity, hi, lo, hist, min_score):
        return self._match(False, node7, submap7, gravity, hi, lo, hist, min_score)

    def match_full_submap(self, node_q, submap_q, gravity, hi, lo, hist, min_score):
        node7 = np.concatenate([[0, 0, 0], node_q])
        submap7 = np.concatenate([[0, 0, 0], submap_q])
        return self._match(True, node7, submap7, gravity, hi, lo, hist, min_score)


# ---- CeresScanMatcher3D: oracle_ceres_3d.h ----
def _ceres3d_args(pairs, occupied_space_weights, translation_weight, rotation_weight,
                  only_optimize_yaw, use_nonmonotonic_steps, max_num_iterations):
    """pairs: [(xyz, resolution, voxels), ...] -- one per (point cloud, hybrid grid)."""
    num = len(pairs)
    assert 1 <= num <= 3 and len(occupied_space_weights) == num
    options = np.zeros(8, np.float64)
    options[:5] = [translation_weight, rotation_weight, 1.0 if only_optimize_yaw else 0.0,
                   1.0 if use_nonmonotonic_steps else 0.0, max_num_iterations]
    options[5:5 + num] = occupied_space_weights
    clouds = [_cloud(p[0])[0] for p in pairs]
    voxels = [_voxels(p[2])[0] for p in pairs]
    keep = (clouds, voxels)
    cloud_ptrs = (C.c_void_p * num)(*[c.ctypes.data for c in clouds])
    counts = np.array([c.shape[0] for c in clouds], np.int32)
    resolutions = np.array([p[1] for p in pairs], np.float32)
    voxel_ptrs = (C.c_void_p * num)(*[v.ctypes.data for v in voxels])
    voxel_counts = np.array([v.shape[0] for v in voxels], np.int64)
    return (options, num, cloud_ptrs, counts.ctypes.data_as(C.c_void_p),
            resolutions.ctypes.data_as(C.c_void_p), voxel_ptrs,
            voxel_counts.ctypes.data_as(C.c_void_p)), (keep, counts, resolutions, voxel_counts)


def ceres3d_match(pairs, target_xyz, init_pose7, occupied_space_weights, translation_weight=5.0,
                  rotation_weight=400.0, only_optimize_yaw=False, use_nonmonotonic_steps=False,
                  max_num_iterations=12, reference=False):
    """CeresScanMatcher3D::Match restated (probability grids only; parity with Ceres's iterates
    is unpinned, see oracle_ceres_3d.h).  init_pose7 = (tx, ty, tz, qw, qx, qy, qz)."""
    args, keep = _ceres3d_args(pairs, occupied_space_weights, translation_weight, rotation_weight,
                               only_optimize_yaw, use_nonmonotonic_steps, max_num_iterations)
    L = lib()
    L.orc_ceres3d_match.argtypes = [_f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, _f64p, _f64p, _f64p, _f64p]
    L.orc_ceres3d_match.restype = None
    pose, summary = np.empty(7, np.float64), np.empty(5, np.float64)
    if reference:
        ref_ceres_lib().refc_ceres3d_match(*args, np.ascontiguousarray(target_xyz, np.float64),
                                           np.ascontiguousarray(init_pose7, np.float64), pose,
                                           summary, None, None, None, None)
    else:
        L.orc_ceres3d_match(*args, np.ascontiguousarray(target_xyz, np.float64),
                            np.ascontiguousarray(init_pose7, np.float64), pose, summary)
    del keep
    return dict(pose=pose, initial_cost=summary[0], final_cost=summary[1],
                num_successful_steps=int(summary[2]), num_unsuccessful_steps=int(summary[3]),
                termination=int(summary[4]))


INTENSITY_VOXEL_DTYPE = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32),
                                  ("count", np.int32), ("sum", np.float32)])


def _intensity_args(pairs):
    """pairs with optional 4th..6th entries (intensities, intensity voxels, (weight, huber_scale,
    threshold)); returns the four trailing arguments of *_ceres3d_match(_intensity) + keepalive."""
    num = len(pairs)
    if not any(len(p) > 3 and p[3] is not None for p in pairs):
        return (None, None, None, None), None
    ints = [np.ascontiguousarray(p[3], np.float32) if len(p) > 3 and p[3] is not None else None
            for p in pairs]
    voxs = [np.ascontiguousarray(p[4], INTENSITY_VOXEL_DTYPE) if ints[i] is not None else None
            for i, p in enumerate(pairs)]
    opts = np.zeros(3 * num, np.float64)
    for i, p in enumerate(pairs):
        if ints[i] is not None:
            opts[3 * i:3 * i + 3] = p[5]
    int_ptrs = (C.c_void_p * num)(*[a.ctypes.data if a is not None else None for a in ints])
    vox_ptrs = (C.c_void_p * num)(*[a.ctypes.data if a is not None else None for a in voxs])
    counts = np.array([a.shape[0] if a is not None else 0 for a in voxs], np.int64)
    return (int_ptrs, vox_ptrs, counts.ctypes.data_as(C.c_void_p),
            opts.ctypes.data_as(C.c_void_p)), (ints, voxs, opts, counts)


def ceres3d_match_intensity(pairs, target_xyz, init_pose7, occupied_space_weights,
                            translation_weight=5.0, rotation_weight=400.0,
                            only_optimize_yaw=False, use_nonmonotonic_steps=False,
                            max_num_iterations=12, reference=False):
    """CeresScanMatcher3D::Match with IntensityCostFunction3D blocks: pairs =
    [(xyz, resolution, voxels, intensities, intensity_voxels, (weight, huber_scale, threshold)), ...]
    (entries 3.. may be missing / None for a pair without intensity grid).  reference=True: the
    reference's own ceres_scan_matcher_3d.cc + intensity_cost_function_3d.{h,cc} (oracle/_ref)."""
    args, keep = _ceres3d_args([p[:3] for p in pairs], occupied_space_weights, translation_weight,
                               rotation_weight, only_optimize_yaw, use_nonmonotonic_steps,
                               max_num_iterations)
    iargs, ikeep = _intensity_args(pairs)
    sig = [_f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _f64p, _f64p,
           _f64p, _f64p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if reference:
        fn = ref_ceres_lib().refc_ceres3d_match
    else:
        fn = lib().orc_ceres3d_match_intensity
        fn.argtypes, fn.restype = sig, None
    pose, summary = np.empty(7, np.float64), np.empty(5, np.float64)
    fn(*args, np.ascontiguousarray(target_xyz, np.float64),
       np.ascontiguousarray(init_pose7, np.float64), pose, summary, *iargs)
    del keep, ikeep
    return dict(pose=pose, initial_cost=summary[0], final_cost=summary[1],
                num_successful_steps=int(summary[2]), num_unsuccessful_steps=int(summary[3]),
                termination=int(summary[4]))


def intensity3d_residuals(scaling_factor, intensity_threshold, xyz, intensities, resolution,
                          intensity_voxels, pose7, reference=False):
    """IntensityCostFunction3D's residuals [n] and Jacobian [n, 7] at pose7 (before the loss)."""
    xyz, n = _cloud(xyz)
    ints = np.ascontiguousarray(intensities, np.float32)
    vox = np.ascontiguousarray(intensity_voxels, INTENSITY_VOXEL_DTYPE)
    fn = ref_ceres_lib().refc_intensity3d_residuals if reference else lib().orc_intensity3d_residuals
    fn.argtypes = [C.c_double, C.c_float, _f32p, _f32p, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                   _f64p, _f64p, _f64p]
    fn.restype = None
    r, J = np.empty(n, np.float64), np.empty((n, 7), np.float64)
    fn(scaling_factor, intensity_threshold, xyz, ints, n, resolution, vox.ctypes.data, vox.shape[0],
       np.ascontiguousarray(pose7, np.float64), r, J)
    return r, J


def ceres3d_residuals(pairs, target_xyz, target_q4, pose7, occupied_space_weights,
                      translation_weight=5.0, rotation_weight=400.0, reference=False):
    """Residuals [N + 6] and their Jacobian [N + 6, 7] w.r.t. (t, q = w x y z) at pose7."""
    args, keep = _ceres3d_args(pairs, occupied_space_weights, translation_weight, rotation_weight,
                               False, False, 0)
    total = int(keep[1].sum()) + 6
    L = lib()
    L.orc_ceres3d_residuals.argtypes = [_f64p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.orc_ceres3d_residuals.restype = None
    r, J = np.empty(total, np.float64), np.empty((total, 7), np.float64)
    fn = ref_ceres_lib().refc_ceres3d_residuals if reference else L.orc_ceres3d_residuals
    fn(*args, np.ascontiguousarray(target_xyz, np.float64),
                            np.ascontiguousarray(target_q4, np.float64),
                            np.ascontiguousarray(pose7, np.float64), r, J)
    del keep
    return r, J


class ReferenceIntensityHybridGrid:
    """The reference's own IntensityHybridGrid (mapping/3d/hybrid_grid.h:543-571); filled through
    ReferenceHybridGrid.insert_with_intensities."""

    def __init__(self, resolution):
        self._h = ref_lib().ref_igrid_create(resolution)

    def __del__(self):
        if getattr(self, "_h", None):
            ref_lib().ref_igrid_destroy(self._h)
            self._h = None

    def voxels(self):
        """Cells with count > 0 as INTENSITY_VOXEL_DTYPE records sorted (z, y, x)."""
        n = ref_lib().ref_igrid_voxels(self._h, None, None, 0)
        rows = np.empty((max(n, 1), 4), np.int32)
        sums = np.empty(max(n, 1), np.float32)
        ref_lib().ref_igrid_voxels(self._h, rows.ctypes.data, sums.ctypes.data, rows.shape[0])
        rows, sums = rows[:n], sums[:n]
        order = np.lexsort((rows[:, 0], rows[:, 1], rows[:, 2]))
        out = np.zeros(n, INTENSITY_VOXEL_DTYPE)
        out["x"], out["y"], out["z"] = rows[order, 0], rows[order, 1], rows[order, 2]
        out["count"], out["sum"] = rows[order, 3], sums[order]
        return out[out["count"] > 0]


def insert_intensities(resolution, voxels, returns_xyz, intensities, intensity_threshold):
    """InsertIntensitiesIntoGrid (range_data_inserter_3d.cc:54-70) restated: `voxels`
    (INTENSITY_VOXEL_DTYPE, may be empty) plus one scan; returns the new voxel list (z, y, x)."""
    ret = np.ascontiguousarray(returns_xyz, np.float32).reshape(-1, 3)
    ints = None if intensities is None else np.ascontiguousarray(intensities, np.float32)
    old = np.ascontiguousarray(voxels, INTENSITY_VOXEL_DTYPE)
    cap = old.shape[0] + ret.shape[0] + 1
    buf = np.zeros(cap, INTENSITY_VOXEL_DTYPE)
    buf[:old.shape[0]] = old
    fn = lib().orc_insert_intensities
    fn.argtypes = [C.c_float, _f32p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int64,
                   C.c_int64]
    fn.restype = C.c_int64
    n = fn(resolution, ret, None if ints is None else ints.ctypes.data, ret.shape[0],
           intensity_threshold, buf.ctypes.data, old.shape[0], cap)
    return buf[:n].copy()
