"""Upstream point preparation: the oracle's voxel filters and rotational
histogram against the reference's own tests and, where oracle/_ref is built, against the
reference's own voxel_filter.cc / rotational_scan_matcher.cc compiled in place."""
import numpy as np
import pytest


def _need_ref(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref is not built here (needs /root/reference)")


def test_voxel_filter_reference_tests(oracle):
    """VoxelFilterTest.ReturnsOnePointInEachVoxel / HandlesLargeCoordinates / IgnoresTime
    (voxel_filter_test.cc:30-84)."""
    cloud = np.array([[0, 0, 0], [0.1, -0.1, 0.1], [0.3, -0.1, 0], [0, 0, 0.1]], np.float32)
    used = oracle.voxel_filter_flags(cloud, 0.3)
    assert used.sum() == 2 and used[2]
    big = np.array([[100000., 0, 0], [100000.001, -0.0001, 0.0001], [100000.003, -0.0001, 0],
                    [-200000., 0, 0]], np.float32)
    used = oracle.voxel_filter_flags(big, 0.01)
    assert used.sum() == 2 and used[3]
    same = np.tile(np.array([[-100.0, 0.3, 0.4]], np.float32), (100, 1))
    assert oracle.voxel_filter_flags(same, 0.3).sum() == 1


@pytest.mark.parametrize("seed,n,res", [(0, 2000, 0.3), (1, 20000, 0.05), (2, 500, 1.0),
                                        (3, 5000, 0.011)])
def test_voxel_filter_equals_the_reference_source(oracle, seed, n, res):
    _need_ref(oracle)
    rng = np.random.default_rng(seed)
    cloud = rng.normal(0.0, 3.0, (n, 3)).astype(np.float32)
    k = len(cloud[1::7])
    cloud[::7][:k] = cloud[1::7]                          # exact duplicates
    used = oracle.voxel_filter_flags(cloud, res)
    ref = oracle.ref_voxel_filter(cloud, res)
    np.testing.assert_array_equal(cloud[used], ref)


@pytest.mark.parametrize("seed,n,max_length,min_points,max_range", [
    (0, 20000, 0.5, 200, 50.0),      # trajectory_builder_2d.lua adaptive_voxel_filter
    (1, 60000, 2.0, 150, 15.0),      # trajectory_builder_3d.lua high resolution
    (2, 60000, 4.0, 200, 60.0),      # low resolution
    (3, 100, 0.5, 200, 50.0),        # already sparse
    (4, 3000, 0.9, 2900, 80.0)])     # needs the binary search down to small voxels
def test_adaptive_voxel_filter_equals_the_reference_source(oracle, seed, n, max_length,
                                                           min_points, max_range):
    _need_ref(oracle)
    rng = np.random.default_rng(seed)
    cloud = (rng.normal(0.0, 8.0, (n, 3)) * np.array([1.0, 1.0, 0.2])).astype(np.float32)
    got = oracle.adaptive_voxel_filter(cloud, max_length, min_points, max_range)
    ref = oracle.ref_adaptive_voxel_filter(cloud, max_length, min_points, max_range)
    np.testing.assert_array_equal(got, ref)
    assert len(got) >= min(min_points, (np.linalg.norm(cloud, axis=1) <= max_range).sum()) or \
        len(got) == len(ref)


@pytest.mark.parametrize("n,size,spread", [(1, 120, 1.0), (2, 7, 0.3), (63, 1, 2.0), (65, 16, 0.5),
                                           (5000, 120, 4.0), (70000, 120, 6.0), (20000, 8192, 3.0)])
def test_compute_histogram_odd_shapes_equal_the_reference_source(oracle, n, size, spread):
    """The shapes the device is checked on (tests/test_gpu_r2_paths.py::
    test_compute_histogram_odd_shapes): the oracle those tests compare with IS the reference's
    rotational_scan_matcher.cc there too -- one-point slices, slices of thousands of points,
    sparse and dense clouds, histogram sizes 1 ... 8192."""
    _need_ref(oracle)
    rng = np.random.default_rng(n + size)
    cloud = (rng.normal(0.0, spread, (n, 3)) * np.array([1.0, 1.0, 0.15])).astype(np.float32)
    np.testing.assert_array_equal(oracle.compute_histogram(cloud, size),
                                  oracle.ref_compute_histogram(cloud, size))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_compute_histogram_equals_the_reference_source(oracle, synth, seed):
    _need_ref(oracle)
    grid, world = synth.make_submap_3d(20 + seed, 0.1, (8.0, 6.0, 3.0), 4, 10, 64)
    pos = world.free_position(seed, 0.5)
    cloud = world.scan(pos, 0.2 * seed, 16, 360, seed=seed)
    got = oracle.compute_histogram(cloud, 120)
    ref = oracle.ref_compute_histogram(cloud, 120)
    np.testing.assert_array_equal(got, ref)
    assert got.sum() > 0
