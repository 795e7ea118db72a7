"""The C++ adapter classes of examples/adapter (reference-shaped FastCorrelativeScanMatcher2D /
RealTimeCorrelativeScanMatcher2D over the C ABI): they compile with g++ against the public
header only, and on a GPU return exactly what the ctypes mirror returns."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "examples", "adapter")


def _build(tmp_path):
    from cartographer_amd import _lib
    exe = str(tmp_path / "adapter_demo")
    lib_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I",
                           os.path.join(ROOT, "include"), "-I", ADAPTER,
                           os.path.join(ADAPTER, "scan_matchers_2d_mi355x.cc"),
                           os.path.join(ADAPTER, "adapter_demo.cc"), "-o", exe, "-L", lib_dir,
                           "-lcartographer_mi355x", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _write_case(path, synth):
    cells, lim, world = synth.make_submap(17, 180, 160, 0.05, 12, 400, 30.0, 0.01)
    truth = world.free_pose(4, 0.5)
    scan = world.scan(truth, 350, 30.0, 0.01, 3)
    init = (truth[0] + 0.1, truth[1] - 0.07, truth[2] + 0.04)
    with open(path, "wb") as f:
        f.write(struct.pack("<3i6d", 180, 160, scan.shape[0], lim["resolution"], lim["max_x"],
                            lim["max_y"], *init))
        f.write(np.ascontiguousarray(cells, np.uint16).tobytes())
        f.write(np.ascontiguousarray(scan, np.float32).tobytes())
    return cells, lim, scan, init


def test_adapter_compiles_and_fails_loudly_without_gpu(tmp_path, synth):
    exe = _build(tmp_path)
    _write_case(str(tmp_path / "case.bin"), synth)
    out = subprocess.run([exe, str(tmp_path / "case.bin")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith(("no device", "full "))


@pytest.mark.gpu
def test_adapter_matches_ctypes_path(tmp_path, synth):
    from cartographer_amd import scan_matching as sm
    exe = _build(tmp_path)
    cells, lim, scan, init = _write_case(str(tmp_path / "case.bin"), synth)
    out = subprocess.run([exe, str(tmp_path / "case.bin")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.strip().splitlines()}
    grid = sm.Grid2D(cells, lim["resolution"], lim["max_x"], lim["max_y"])
    fast = sm.FastCorrelativeScanMatcher2D(grid, 6, 7.0, 0.5)
    for key, (found, score, pose) in (("full", fast.match_full_submap(scan, 0.5)),
                                      ("window", fast.match(sm.Rigid2d(*init), scan, 0.5))):
        assert found and int(lines[key][0]) == 1
        assert np.float32(lines[key][1]) == np.float32(score)
        assert [float(x) for x in lines[key][2:]] == [pose.x, pose.y, pose.theta]
    assert int(lines["none"][0]) == 0
    rt = sm.RealTimeCorrelativeScanMatcher2D(0.3, 0.12, 0.1, 0.1)
    score, pose = rt.match(sm.Rigid2d(*init), scan, grid)
    assert [float(x) for x in lines["rt"]] == [score, pose.x, pose.y, pose.theta]


def _build_3d(tmp_path):
    from cartographer_amd import _lib
    exe = str(tmp_path / "adapter_demo_3d")
    lib_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I",
                           os.path.join(ROOT, "include"), "-I", ADAPTER,
                           os.path.join(ADAPTER, "scan_matchers_3d_mi355x.cc"),
                           os.path.join(ADAPTER, "adapter_demo_3d.cc"), "-o", exe, "-L", lib_dir,
                           "-lcartographer_mi355x", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _write_case_3d(path, synth):
    grid, world = synth.make_submap_3d(11, 0.1, (6.0, 5.0, 3.0), 3, 8, 64)
    vox = grid.voxels()
    hist = np.zeros(8, np.float32)
    pos = world.free_position(3, 0.5)
    hi = world.scan(pos, 0.0, 6, 48, seed=4)
    lo = hi[::5].copy()
    node = [pos[0] + 0.2, pos[1] - 0.1, pos[2], 1.0, 0.0, 0.0, 0.0]
    with open(path, "wb") as f:
        f.write(struct.pack("<5if7d", len(vox), grid.grid_size, hi.shape[0], lo.shape[0], 8, 0.1,
                            *node))
        f.write(np.ascontiguousarray(vox).tobytes())
        f.write(np.ascontiguousarray(hi, np.float32).tobytes())
        f.write(np.ascontiguousarray(lo, np.float32).tobytes())
        f.write(hist.tobytes())
    return grid, vox, hist, hi, lo, node


def test_adapter_3d_compiles_and_fails_loudly_without_gpu(tmp_path, synth):
    exe = _build_3d(tmp_path)
    _write_case_3d(str(tmp_path / "case3d.bin"), synth)
    out = subprocess.run([exe, str(tmp_path / "case3d.bin")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith(("no device", "fast "))


@pytest.mark.gpu
def test_adapter_3d_matches_ctypes_path(tmp_path, synth):
    from cartographer_amd import scan_matching_3d as sm3
    exe = _build_3d(tmp_path)
    grid, vox, hist, hi, lo, node = _write_case_3d(str(tmp_path / "case3d.bin"), synth)
    out = subprocess.run([exe, str(tmp_path / "case3d.bin")], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.strip().splitlines()}
    opts = dict(branch_and_bound_depth=5, full_resolution_depth=2, min_rotational_score=0.5,
                min_low_resolution_score=0.25, linear_xy_search_window=1.0,
                linear_z_search_window=0.5, angular_search_window=0.1)
    f3 = sm3.FastCorrelativeScanMatcher3D(0.1, vox, grid.grid_size, 0.1, vox, hist, **opts)
    pose = sm3.Rigid3d(tuple(node[:3]), tuple(node[3:]))
    got = f3.match(pose, sm3.Rigid3d(), sm3.TrajectoryNodeData(hi, lo, hist), 0.3)
    assert (got is not None) == (lines["fast"][0] == "1")
    if got is not None:
        vals = [float(x) for x in lines["fast"][1:]]
        assert np.float32(vals[0]) == np.float32(got["score"])
        assert np.float32(vals[1]) == np.float32(got["rotational_score"])
        assert np.float32(vals[2]) == np.float32(got["low_resolution_score"])
        assert vals[3:] == list(got["pose_estimate"].as_array())
    assert lines["none"] == ["0"]
    rt = sm3.RealTimeCorrelativeScanMatcher3D(0.1, 0.02, 0.1, 0.1)
    score, est = rt.match(pose, hi, 0.1, vox)
    vals = [float(x) for x in lines["rt"]]
    assert np.float32(vals[0]) == np.float32(score)
    assert vals[1:] == list(est.as_array())
