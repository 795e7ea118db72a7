#!/usr/bin/env python3
"""Generates tests/golden/rt2d_tsdf_fixture.npz: the TSDF2D of the reference's own real-time
matcher test (RealTimeCorrelativeScanMatcherTest::SetUpTSDF,
mapping/internal/2d/scan_matching/real_time_correlative_scan_matcher_2d_test.cc:66-92), built by
the REFERENCE'S OWN TSDFRangeDataInserter2D / normal_estimation_2d / TSDF2D sources (compiled
unmodified by `make -C oracle ref` against the stand-in headers of oracle/ref_shims; the stand-in
Eigen value types evaluate 2- and 3-term norms / dot products left to right).

Needs /root/reference (or a prebuilt oracle/_ref/libref.so); the fixture is committed so that the
GPU box, which has neither requirement guaranteed, can use it.  Usage:
    python tests/golden/make_tsdf_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

# The 7-point L-shaped cloud of the test fixture (:55-61).
L_CLOUD = np.array([[0.025, 0.175, 0], [-0.025, 0.175, 0], [-0.075, 0.175, 0],
                    [-0.125, 0.175, 0], [-0.125, 0.125, 0], [-0.125, 0.075, 0],
                    [-0.125, 0.025, 0]], np.float32)
TRUNCATION_DISTANCE, MAX_WEIGHT = 0.3, 1.0        # TSDF2D(..., 0.3, 1.0, ...), :67-69


def build():
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        raise SystemExit("oracle/_ref is not built and /root/reference is absent")
    grid = orc.ReferenceTSDF2D(0.05, (0.3, 0.5), 20, 20, TRUNCATION_DISTANCE, MAX_WEIGHT)
    # the dictionary of :70-84; origin (0.5, -0.5, 0), :86-88
    grid.insert([0.5, -0.5, 0.0], L_CLOUD, truncation_distance=0.3, maximum_weight=10.0,
                update_free_space=False, num_normal_samples=4, sample_radius=0.5,
                project_sdf_distance_to_scan_normal=True, update_weight_range_exponent=0,
                angle_kernel_bandwidth=0.5, distance_kernel_bandwidth=0.5)
    tsd, wgt = grid.planes()
    lim = grid.limits
    return dict(tsd=tsd, weight=wgt, resolution=lim["resolution"], max_x=lim["max_x"],
                max_y=lim["max_y"], truncation_distance=TRUNCATION_DISTANCE,
                max_weight=MAX_WEIGHT, cloud=L_CLOUD)


if __name__ == "__main__":
    out = os.path.join(HERE, "rt2d_tsdf_fixture.npz")
    np.savez_compressed(out, **build())
    print("wrote", out)
