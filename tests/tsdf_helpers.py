"""TSDF2D fixtures for the tests: tsd / weight planes filled through the restated
TSDValueConverter (oracle), the way TSDF2D::SetCell stores them
(mapping/internal/2d/tsdf_2d.cc:50-66).

The reference's own TSDF fixtures go through TSDFRangeDataInserter2D (normal
estimation + ray casting), which is outside the scan-matching path and is not
restated; these helpers build the distance field of the same scene directly.
"""
import numpy as np


def cell_centres(res, max_x, max_y, nx, ny):
    """MapLimits::GetCellCenter (mapping/2d/map_limits.h:78-83) for cells[iy][ix]."""
    iy, ix = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    return max_x - res * (iy + 0.5), max_y - res * (ix + 0.5)


def _distance_to_polyline(px, py, poly):
    best = np.full(px.shape, np.inf)
    for (ax, ay), (bx, by) in zip(poly[:-1], poly[1:]):
        dx, dy = bx - ax, by - ay
        t = np.clip(((px - ax) * dx + (py - ay) * dy) / (dx * dx + dy * dy), 0.0, 1.0)
        best = np.minimum(best, np.hypot(px - (ax + t * dx), py - (ay + t * dy)))
    return best


def planes_from_distance(orc, distance, known, weight, truncation, max_weight):
    """uint16 (tsd, weight) planes from a per-cell distance / weight; unknown cells stay 0."""
    tsd = np.zeros(distance.shape, np.uint16)
    wgt = np.zeros(distance.shape, np.uint16)
    for iy, ix in zip(*np.nonzero(known)):
        tsd[iy, ix] = orc.tsd_to_value(float(distance[iy, ix]), truncation)
        wgt[iy, ix] = orc.weight_to_value(float(weight[iy, ix]), max_weight)
    return tsd, wgt


def polyline_tsdf(orc, poly, res, max_x, max_y, nx, ny, truncation, max_weight, weight=1.0):
    cx, cy = cell_centres(res, max_x, max_y, nx, ny)
    d = _distance_to_polyline(cx, cy, poly)
    known = d < truncation
    return planes_from_distance(orc, d, known, np.full(d.shape, weight), truncation, max_weight)


def tsdf_from_probability_grid(orc, cells, res, truncation, max_weight, seed):
    """A TSDF of the scene a probability grid shows: unsigned distance to the nearest
    occupied cell, random weights, a band of unknown cells, some update markers."""
    from scipy import ndimage
    occupied = (cells & 32767) > 0
    prob = np.array([[orc.grid_probability(cells, ix, iy) for ix in range(cells.shape[1])]
                     for iy in range(cells.shape[0])]) if cells.size <= 4096 else None
    if prob is None:
        # kValueToProbability is monotone in the raw value: occupied <=> cost below 0.5,
        # i.e. raw correspondence-cost value in the lower half of [1, 32767].
        wall = occupied & ((cells & 32767) < 16384)
    else:
        wall = occupied & (prob > 0.5)
    d = ndimage.distance_transform_edt(~wall) * res
    rng = np.random.default_rng(seed)
    weight = rng.uniform(0.0, max_weight, cells.shape)
    known = (d < truncation) & (rng.uniform(size=cells.shape) > 0.05)
    sign = np.where(rng.uniform(size=cells.shape) < 0.5, -1.0, 1.0)
    tsd, wgt = planes_from_distance(orc, d * sign, known, weight, truncation, max_weight)
    marked = known & (rng.uniform(size=cells.shape) < 0.1)
    tsd[marked] |= 1 << 15          # update marker (tsdf_2d.cc:60-61) must be ignored
    return tsd, wgt
