"""The C++ drop-in proof: the reference's ConstraintBuilder2D, compiled unmodified, with the
MI355X matchers underneath (examples/dropin).

CPU: the executable builds from the reference tree when it is present, links the product
library and nothing of the oracle.  GPU: the prebuilt executable runs the reference's own
CallsBack / FindsConstraints scenario (constraint_builder_2d_test.cc:58-112) and a realistic
node whose constraints must equal what the oracle computes for ComputeConstraint
(constraint_builder_2d.cc:204-282): fast correlative match (local window or full submap), Ceres
refinement from that pose, then submap_pose^-1 * pose_estimate.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "examples", "dropin")
BINARY = os.path.join(DROPIN, "_build", "constraint_builder_2d_mi355x")
REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_builds_from_the_reference_tree_and_links_only_the_product():
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    needed = subprocess.run(["readelf", "-d", BINARY], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed
    assert "oracle" not in needed
    # the reference's sources are compiled where they lie: none of them lives in the repo
    for name in ("constraint_builder_2d.cc", "thread_pool.cc", "task.cc"):
        assert not os.path.exists(os.path.join(DROPIN, name))


def _write_fixture(path, submaps, scan, rel):
    nx, ny, res = 400, 400, 0.05
    with open(path, "wb") as f:
        f.write(struct.pack("<iiid", len(submaps), nx, ny, res))
        for cells, lim, origin in submaps:
            f.write(struct.pack("<dddd", lim["max_x"], lim["max_y"], origin[0], origin[1]))
            f.write(np.ascontiguousarray(cells, np.uint16).tobytes())
        f.write(struct.pack("<i", len(scan)))
        f.write(np.ascontiguousarray(scan, np.float32).tobytes())
        f.write(struct.pack("<ddd", *rel))


@pytest.mark.gpu
def test_reference_constraint_builder_on_the_gpu(oracle, synth, tmp_path):
    assert os.path.exists(BINARY), "examples/dropin/_build is prebuilt by __graft_entry__.build()"
    truth_world = None
    submaps = []
    for seed, origin in ((42, (1.5, -2.25)), (43, (0.0, 0.0)), (42, (-3.0, 4.5))):
        cells, lim, world = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        if truth_world is None:
            truth_world = world
        submaps.append((cells, lim, origin))
    truth = truth_world.free_pose(1234, 0.5)
    scan = truth_world.scan(truth, 1000, 30.0, 0.01, 7)
    # the pose graph's guess of the node in submap 0's frame: truth off by a few decimetres
    rel = (truth[0] - 1.5 + 0.35, truth[1] + 2.25 - 0.25, truth[2] + 0.12)
    fixture = str(tmp_path / "node.bin")
    _write_fixture(fixture, submaps, scan, rel)
    out = subprocess.run([BINARY, fixture], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    assert "reference scenario: CallsBack + FindsConstraints OK" in out
    got = []
    for line in out.splitlines():
        if line.startswith("constraint submap"):
            w = line.split()
            got.append((int(w[2]), float(w[6]), float(w[7]), float(w[9]), int(w[11])))
    # ComputeConstraint, restated with the oracle.
    want = []
    for k, (cells, lim, origin) in enumerate(submaps):
        fast = oracle.FastCorrelativeScanMatcher2D(cells, 0.05, lim["max_x"], lim["max_y"], 7,
                                                   7.0, np.radians(30.0))
        init = [origin[0] + rel[0], origin[1] + rel[1], rel[2]]
        for res in (fast.match(init, scan, 0.55), fast.match_full_submap(scan, 0.6)):
            if not res["found"]:
                continue
            p = res["pose"]
            refined = oracle.ceres2d_match(cells, 0.05, lim["max_x"], lim["max_y"], p[:2], p,
                                           scan, 20.0, 10.0, 1.0, True, 10)["pose"]
            want.append((k, refined[0] - origin[0], refined[1] - origin[1], refined[2]))
    assert len(want) >= 3, "the scenario should close loops on both copies of the submap"
    assert f"constraints {len(want)}" in out
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[4] == 1          # INTER_SUBMAP
        np.testing.assert_allclose(g[1:4], w[1:4], rtol=0, atol=1e-6)
        # and the loop closure is right: node pose in the submap frame = truth - origin (the
        # two copies of the scan's own world; submap 1 is another world, whatever matches there
        # is a false positive the reference would report just the same)
        if g[0] == 1:
            continue
        origin = submaps[g[0]][2]
        assert abs(g[1] - (truth[0] - origin[0])) < 0.05
        assert abs(g[2] - (truth[1] - origin[1])) < 0.05
        assert abs(g[3] - truth[2]) < 0.01


# ------------------------------------------------------------------------------------- 3D
BINARY_3D = os.path.join(DROPIN, "_build", "constraint_builder_3d_mi355x")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_3d_builds_from_the_reference_tree_and_links_only_the_product():
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    needed = subprocess.run(["readelf", "-d", BINARY_3D], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed
    assert "oracle" not in needed
    for name in ("constraint_builder_3d.cc", "hybrid_grid.h", "thread_pool.cc"):
        assert not os.path.exists(os.path.join(DROPIN, name))


def _write_fixture_3d(path, options, resolutions, submaps, hi, lo, hist, node7):
    def grid(f, voxels):
        f.write(struct.pack("<q", len(voxels)))
        f.write(np.ascontiguousarray(voxels).tobytes())          # x, y, z int32; value, pad u16

    def cloud(f, xyz):
        f.write(struct.pack("<i", len(xyz)))
        f.write(np.ascontiguousarray(xyz, np.float32).tobytes())

    def histogram(f, h):
        f.write(struct.pack("<i", len(h)))
        f.write(np.ascontiguousarray(h, np.float32).tobytes())

    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(submaps)))
        f.write(struct.pack("<9d", *options))
        f.write(struct.pack("<2f", *resolutions))
        for vox, low_vox, h in submaps:
            histogram(f, h)
            grid(f, vox)
            grid(f, low_vox)
        cloud(f, hi)
        cloud(f, lo)
        histogram(f, hist)
        f.write(struct.pack("<7d", *node7))


@pytest.mark.gpu
def test_reference_constraint_builder_3d_on_the_gpu(oracle, synth, tmp_path):
    """The reference's ConstraintBuilder3D (unmodified source) on the device: its own test
    scenario (an EMPTY submap, one point, thresholds 0: constraint_builder_3d_test.cc:61-122)
    and a node against three submaps whose constraints must equal ComputeConstraint restated
    with the oracle (constraint_builder_3d.cc:188-281): Match / MatchFullSubmap, then
    CeresScanMatcher3D::Match from that pose with the high- and low-resolution pairs."""
    import math
    assert os.path.exists(BINARY_3D), "examples/dropin/_build is prebuilt by __graft_entry__.build()"
    hist = np.zeros(16, np.float32)
    submaps, world = [], None
    for seed in (31, 32, 31):
        g, w = synth.make_submap_3d(seed, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
        low, _ = synth.make_submap_3d(seed, 0.4, (8.0, 8.0, 3.0), 4, 8, 96)
        world = world or w
        submaps.append((g.voxels(), low.voxels(), hist))
    pos = world.free_position(5, 0.6)
    hi = world.scan(pos, 0.0, 6, 64, seed=2)
    lo = hi[::5].copy()
    yaw = 0.03
    node7 = list(pos + np.array([0.15, -0.1, 0.05])) + [math.cos(yaw / 2), 0.0, 0.0,
                                                        math.sin(yaw / 2)]
    # min_score, global min_score, depth, full_resolution_depth, min_rotational_score,
    # min_low_resolution_score, xy / z / angular windows
    options = (0.4, 0.4, 5, 2, 0.5, 0.25, 1.0, 1.0, 0.1)
    fixture = str(tmp_path / "node3d.bin")
    _write_fixture_3d(fixture, options, (0.2, 0.4), submaps, hi, lo, hist, node7)
    out = subprocess.run([BINARY_3D, fixture], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    assert "reference scenario: CallsBack + FindsConstraints OK" in out
    got = []
    for line in out.splitlines():
        if line.startswith("constraint submap"):
            w = line.split()
            got.append((int(w[2]), [float(v) for v in w[6:9]] + [float(v) for v in w[10:14]],
                        int(w[15])))
    want = []
    ident7 = [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]
    for k, (vox, low_vox, h) in enumerate(submaps):
        fast = oracle.FastCorrelativeScanMatcher3D(0.2, vox, 0.4, low_vox, h, 5, 2, 0.5, 0.25, 1.0,
                                                   1.0, 0.1)
        pairs = [(hi, 0.2, vox), (lo, 0.4, low_vox)]
        for res in (fast.match(node7, ident7, (1, 0, 0, 0), hi, lo, hist, 0.4),
                    fast.match_full_submap(node7[3:], [1, 0, 0, 0], (1, 0, 0, 0), hi, lo, hist,
                                           0.4)):
            if not res["found"]:
                continue
            refined = oracle.ceres3d_match(pairs, res["pose"][:3], list(res["pose"]), [5.0, 30.0],
                                           translation_weight=10.0, rotation_weight=1.0,
                                           max_num_iterations=10)["pose"]
            want.append((k, refined))
    assert len(want) == 5, "local matches in both copies of the world, global ones in all three"
    assert f"constraints {len(want)}" in out
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[2] == 1          # INTER_SUBMAP
        np.testing.assert_allclose(g[1], w[1], rtol=0, atol=1e-6)


# ------------------------------------------------------------------- batched C++ front, 2D
BINARY_BATCHED = os.path.join(DROPIN, "_build", "constraint_builder_2d_batched_mi355x")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_batched_front_builds_with_the_reference_interface():
    """examples/dropin/batched: a ConstraintBuilder2D with the reference's public interface whose
    NotifyEndOfNode hands a node's pairs to ONE cmx_fast2d_match_sharded over the builder's
    cmx_comm (CMX_DEVICES; submap k on device k mod world) + cmx_fast2d_refine_batch.  The
    reference's own test main compiles against it unchanged (include redirect)."""
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    needed = subprocess.run(["readelf", "-d", BINARY_BATCHED], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed and "oracle" not in needed
    symbols = subprocess.run(["nm", "-C", BINARY_BATCHED], check=True, capture_output=True,
                             text=True).stdout
    assert "cmx_fast2d_match_sharded" in symbols and "cmx_fast2d_refine_batch" in symbols
    assert "cmx_comm_init" in symbols and "cmx_comm_device_of" in symbols
    # the per-pair adapter classes are not part of this build
    assert "FastCorrelativeScanMatcher2D::Match" not in symbols


@pytest.mark.gpu
def test_batched_front_gives_the_reference_builders_constraints(synth, tmp_path):
    """The batched builder and the reference's unmodified builder (over the per-pair adapters)
    run the same main on the same fixture: the reference's test scenario passes and the printed
    constraints are identical, digit for digit."""
    assert os.path.exists(BINARY) and os.path.exists(BINARY_BATCHED)
    truth_world = None
    submaps = []
    for seed, origin in ((42, (1.5, -2.25)), (43, (0.0, 0.0)), (42, (-3.0, 4.5))):
        cells, lim, world = synth.make_submap(seed, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
        truth_world = truth_world or world
        submaps.append((cells, lim, origin))
    truth = truth_world.free_pose(1234, 0.5)
    scan = truth_world.scan(truth, 1000, 30.0, 0.01, 7)
    rel = (truth[0] - 1.5 + 0.35, truth[1] + 2.25 - 0.25, truth[2] + 0.12)
    fixture = str(tmp_path / "node.bin")
    _write_fixture(fixture, submaps, scan, rel)
    outs = []
    # the batched builder twice: a 1-device communicator, and every GPU of the box ("all": the
    # submaps then live round-robin on all of them; on a 1-GPU box the same as the first)
    for binary, devices in ((BINARY, None), (BINARY_BATCHED, "0"), (BINARY_BATCHED, "all")):
        env = dict(os.environ)
        env.pop("CMX_DEVICES", None)
        if devices:
            env["CMX_DEVICES"] = devices
        out = subprocess.run([binary, fixture], check=True, capture_output=True, text=True,
                             timeout=300, env=env).stdout
        assert "reference scenario: CallsBack + FindsConstraints OK" in out
        outs.append([line for line in out.splitlines() if line.startswith("constraint")])
    assert len(outs[0]) >= 4 and outs[0] == outs[1] == outs[2]


# ------------------------------------------------------------------- batched C++ front, 3D
BINARY_BATCHED_3D = os.path.join(DROPIN, "_build", "constraint_builder_3d_batched_mi355x")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_batched_front_3d_builds_with_the_reference_interface():
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    needed = subprocess.run(["readelf", "-d", BINARY_BATCHED_3D], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed and "oracle" not in needed
    symbols = subprocess.run(["nm", "-C", BINARY_BATCHED_3D], check=True, capture_output=True,
                             text=True).stdout
    assert "cmx_fast3d_match_sharded" in symbols and "cmx_fast3d_refine_batch" in symbols
    assert "cmx_comm_init" in symbols and "cmx_comm_device_of" in symbols
    assert "FastCorrelativeScanMatcher3D::Match" not in symbols


@pytest.mark.gpu
def test_batched_front_3d_gives_the_reference_builders_constraints(synth, tmp_path):
    """The batched ConstraintBuilder3D and the reference's unmodified one run the same main on
    the same fixture: identical constraints, digit for digit."""
    import math
    assert os.path.exists(BINARY_3D) and os.path.exists(BINARY_BATCHED_3D)
    hist = np.zeros(16, np.float32)
    submaps, world = [], None
    for seed in (31, 32, 31):
        g, w = synth.make_submap_3d(seed, 0.2, (8.0, 8.0, 3.0), 4, 8, 96)
        low, _ = synth.make_submap_3d(seed, 0.4, (8.0, 8.0, 3.0), 4, 8, 96)
        world = world or w
        submaps.append((g.voxels(), low.voxels(), hist))
    pos = world.free_position(5, 0.6)
    hi = world.scan(pos, 0.0, 6, 64, seed=2)
    lo = hi[::5].copy()
    node7 = list(pos + np.array([0.15, -0.1, 0.05])) + [math.cos(0.015), 0.0, 0.0, math.sin(0.015)]
    fixture = str(tmp_path / "node3d.bin")
    _write_fixture_3d(fixture, (0.4, 0.4, 5, 2, 0.5, 0.25, 1.0, 1.0, 0.1), (0.2, 0.4), submaps, hi,
                      lo, hist, node7)
    outs = []
    for binary, devices in ((BINARY_3D, None), (BINARY_BATCHED_3D, "0"), (BINARY_BATCHED_3D, "all")):
        env = dict(os.environ)
        env.pop("CMX_DEVICES", None)
        if devices:                      # the builder's cmx_comm: one device, then all of the box
            env["CMX_DEVICES"] = devices
        out = subprocess.run([binary, fixture], check=True, capture_output=True, text=True,
                             timeout=300, env=env).stdout
        assert "reference scenario: CallsBack + FindsConstraints OK" in out
        outs.append([line for line in out.splitlines() if line.startswith("constraint")])
    assert len(outs[0]) == 6 and outs[0] == outs[1] == outs[2]   # 5 constraints + the count line


# ---- the real-time matchers against the reference's REAL class headers -------------------------
RT_REFERENCE = os.path.join(DROPIN, "_build", "real_time_matchers_reference")
RT_MI355X = os.path.join(DROPIN, "_build", "real_time_matchers_mi355x")
RT_GOLDEN = os.path.join(ROOT, "tests", "golden", "real_time_matchers_reference.txt")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_real_time_matchers_build_against_the_real_headers_and_the_golden_is_the_references():
    """real_time_matchers_main.cc (the scenarios of real_time_correlative_scan_matcher_2d_test.cc
    and _3d_test.cc) links twice: over the reference's own .cc files -- that binary runs here and
    must print the committed golden, every number a hex float -- and over
    real_time_matchers_mi355x.cc, which includes the reference's real class headers (no stand-in
    of ours for them) and links the product library only."""
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    out = subprocess.run([RT_REFERENCE], check=True, capture_output=True, text=True,
                         timeout=120).stdout
    assert out == open(RT_GOLDEN).read()
    assert out.count("\n") == 24 and "x4 + RealTimeCorrelativeScanMatcher3DTest x7 OK" in out
    needed = subprocess.run(["readelf", "-d", RT_MI355X], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed and "oracle" not in needed
    # the adapter names the reference's headers, not stand-ins: neither shim directory has them
    for shim_root in (os.path.join(DROPIN, "shims"), os.path.join(ROOT, "oracle", "ref_shims")):
        for name in ("real_time_correlative_scan_matcher_2d.h",
                     "real_time_correlative_scan_matcher_3d.h"):
            assert not any(name in files for _, _, files in os.walk(shim_root))
    # without a GPU the adapter fails loudly instead of computing anything on the host
    if not os.path.exists("/dev/kfd"):
        run = subprocess.run([RT_MI355X], capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and "no CPU fallback" in run.stderr


@pytest.mark.gpu
def test_real_time_matchers_on_the_gpu_print_the_references_numbers():
    """The same scenarios on the device: every expectation of the reference's two test files
    holds (the binary exits non-zero otherwise) and every score and pose -- ScoreCandidates on a
    ProbabilityGrid and a TSDF2D built by the reference's own inserters, Match() on both, the
    seven 3D cases -- is bit-identical to what the reference's own sources print."""
    assert os.path.exists(RT_MI355X), "examples/dropin/_build is prebuilt by __graft_entry__.build()"
    out = subprocess.run([RT_MI355X], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    assert out.splitlines() == open(RT_GOLDEN).read().splitlines()


# ---- the reference's LocalTrajectoryBuilder2D, unmodified, scan after scan ---------------------
LTB_REFERENCE = os.path.join(DROPIN, "_build", "local_trajectory_builder_2d_reference")
LTB_MI355X = os.path.join(DROPIN, "_build", "local_trajectory_builder_2d_mi355x")
LTB_GOLDEN = os.path.join(ROOT, "tests", "golden", "local_trajectory_builder_2d_reference.txt")


def _drive(text):
    """(scan index, [x, y, yaw], points matched, submaps inserted into) of every result line,
    the submap digests, the worst distance from the simulated truth."""
    poses, submaps, worst = [], [], None
    for line in text.splitlines():
        w = line.split()
        if line.startswith("scan") and "pose" in line:
            poses.append((int(w[1]), [float(v) for v in w[5:8]], int(w[13]), int(w[15])))
        elif line.startswith("submap"):
            submaps.append((int(w[2]), int(w[4]), int(w[6]), int(w[8]), int(w[10]), float(w[12])))
        elif line.startswith("results"):
            worst = float(w[5])
    return poses, submaps, worst


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_local_trajectory_builder_2d_builds_unmodified_and_the_golden_is_the_references():
    """local_trajectory_builder_2d.cc compiles where it lies, twice: with the reference's own
    real-time matcher, Ceres matcher and voxel filter (that binary runs here; its drive of 80
    simulated scans must print the committed golden and stay within 2.5 cm of the truth) and
    with the three adapter files over the product library only."""
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    out = subprocess.run([LTB_REFERENCE], check=True, capture_output=True, text=True,
                         timeout=120).stdout
    assert out == open(LTB_GOLDEN).read()
    poses, submaps, worst = _drive(out)
    assert len(poses) == 79 and worst < 0.025
    assert sum(1 for p in poses if p[3] == 2) > 20      # a second submap was started and filled
    assert len(submaps) == 2
    needed = subprocess.run(["readelf", "-d", LTB_MI355X], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed and "oracle" not in needed
    symbols = subprocess.run(["nm", "-C", LTB_MI355X], check=True, capture_output=True,
                             text=True).stdout
    for name in ("cmx_rt2d_match", "cmx_ceres2d_match", "cmx_voxel_filter",
                 "cmx_adaptive_voxel_filter"):
        assert name in symbols
    assert "LocalTrajectoryBuilder2D::AddAccumulatedRangeData" in symbols
    # neither the builder nor its header is stood in or copied
    for root in (DROPIN, os.path.join(ROOT, "oracle", "ref_shims")):
        for _, _, files in os.walk(root):
            assert "local_trajectory_builder_2d.cc" not in files
            assert "local_trajectory_builder_2d.h" not in files
    if not os.path.exists("/dev/kfd"):
        run = subprocess.run([LTB_MI355X], capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and "no CPU fallback" in run.stderr


LTB_RESIDENT = os.path.join(DROPIN, "_build", "local_trajectory_builder_2d_resident_mi355x")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_resident_local_trajectory_builders_link_the_device_grid_calls():
    """The third build of both drives (examples/dropin/resident): the submaps' grids live in HBM
    -- insertion, crop and both matchers through the handle entry points -- and the reference's
    range-data inserters are not even linked."""
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    symbols = subprocess.run(["nm", "-C", LTB_RESIDENT], check=True, capture_output=True,
                             text=True).stdout
    for name in ("cmx_grid2d_create", "cmx_grid2d_insert", "cmx_grid2d_crop",
                 "cmx_rt2d_match_grid", "cmx_ceres2d_match_grid"):
        assert name in symbols
    assert "ProbabilityGridRangeDataInserter2D::Insert" not in symbols
    assert "LocalTrajectoryBuilder2D::AddAccumulatedRangeData" in symbols
    symbols = subprocess.run(["nm", "-C", LTB3_RESIDENT], check=True, capture_output=True,
                             text=True).stdout
    for name in ("cmx_grid3d_create", "cmx_grid3d_insert", "cmx_rt3d_match_grid",
                 "cmx_ceres3d_match_grids"):
        assert name in symbols
    assert "RangeDataInserter3D::Insert" not in symbols
    assert "LocalTrajectoryBuilder3D::AddAccumulatedRangeData" in symbols


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["adapters", "resident"])
def test_local_trajectory_builder_2d_on_the_gpu_follows_the_references_drive(binary):
    """The same 80 scans with the device under the unmodified builder ("resident": the submaps'
    grids in HBM too, inserted into by cmx_grid2d_insert): voxel filters
    (bit-exact), real-time correlative matcher (bit-exact), Ceres matcher (the device's solver:
    equal to ~1e-9 per solve).  The loop is closed -- every pose moves the next scan's insertion
    -- so the rounding-level differences of the solver grow once they flip a grid cell: the first
    ten results must agree to 1e-6, all of them to 5 mm, the same scans must be inserted into the
    same number of submaps, and the drive must stay as close to the truth as the reference's."""
    path = LTB_MI355X if binary == "adapters" else LTB_RESIDENT
    assert os.path.exists(path), "examples/dropin/_build is prebuilt by __graft_entry__.build()"
    out = subprocess.run([path], check=True, capture_output=True, text=True, timeout=300).stdout
    got, got_submaps, got_worst = _drive(out)
    want, want_submaps, want_worst = _drive(open(LTB_GOLDEN).read())
    assert len(got) == len(want) == 79
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[0] == w[0] and g[3] == w[3]
        np.testing.assert_allclose(g[1], w[1], rtol=0, atol=1e-6 if k < 10 else 5e-3)
        assert abs(g[2] - w[2]) <= 3                   # points the adaptive filter kept
    assert got_worst < 0.025 and abs(got_worst - want_worst) < 5e-3
    for g, w in zip(got_submaps, want_submaps):
        assert g[:4] == w[:4]                          # scans, finished, cells
        assert abs(g[4] - w[4]) < 0.01 * w[4] and abs(g[5] - w[5]) < 0.01 * w[5]


# ---- the reference's LocalTrajectoryBuilder3D, unmodified, scan after scan ---------------------
LTB3_REFERENCE = os.path.join(DROPIN, "_build", "local_trajectory_builder_3d_reference")
LTB3_MI355X = os.path.join(DROPIN, "_build", "local_trajectory_builder_3d_mi355x")
LTB3_GOLDEN = os.path.join(ROOT, "tests", "golden", "local_trajectory_builder_3d_reference.txt")


def _drive_3d(text):
    """(scan index, [x, y, z, qw, qx, qy, qz], high- and low-resolution points, submaps inserted
    into) per result line, (scans, finished, histogram sum) and (voxels, value sum) x 2 per
    submap, the worst distance from the simulated truth."""
    poses, submaps, worst = [], [], None
    for line in text.splitlines():
        w = line.split()
        if line.startswith("scan") and "pose" in line:
            poses.append((int(w[1]), [float(v) for v in w[5:8] + w[9:13]], int(w[18]), int(w[19]),
                          int(w[21])))
        elif line.startswith("submap"):
            submaps.append([int(w[2]), int(w[4]), float(w[6])])
        elif line.startswith("  high") or line.startswith("  low"):
            submaps[-1] += [int(w[4]), int(w[6])]
        elif line.startswith("results"):
            worst = float(w[5])
    return poses, submaps, worst


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree to compile")
def test_local_trajectory_builder_3d_builds_unmodified_and_the_golden_is_the_references():
    """local_trajectory_builder_3d.cc compiles where it lies, twice: with the reference's own
    real-time matcher, Ceres matcher and voxel filter (that binary runs here and must print the
    committed golden: 60 simulated 16-beam sweeps with an IMU) and with the three adapter files
    over the product library only."""
    subprocess.run(["make", "-C", DROPIN], check=True, capture_output=True)
    out = subprocess.run([LTB3_REFERENCE], check=True, capture_output=True, text=True,
                         timeout=300).stdout
    assert out == open(LTB3_GOLDEN).read()
    poses, submaps, worst = _drive_3d(out)
    # (the reference's matcher underestimates this drive's motion by about a tenth at 0.1 m voxels
    # and 400 matched points; what is pinned here is that the device build does exactly the same)
    assert len(poses) == 60 and worst < 0.25
    assert sum(1 for p in poses if p[4] == 2) > 20 and len(submaps) == 2
    needed = subprocess.run(["readelf", "-d", LTB3_MI355X], check=True, capture_output=True,
                            text=True).stdout
    assert "libcartographer_mi355x.so" in needed and "oracle" not in needed
    symbols = subprocess.run(["nm", "-C", LTB3_MI355X], check=True, capture_output=True,
                             text=True).stdout
    for name in ("cmx_rt3d_match", "cmx_ceres3d_match", "cmx_voxel_filter",
                 "cmx_adaptive_voxel_filter"):
        assert name in symbols
    assert "LocalTrajectoryBuilder3D::AddAccumulatedRangeData" in symbols
    for root in (DROPIN, os.path.join(ROOT, "oracle", "ref_shims")):
        for _, _, files in os.walk(root):
            assert "local_trajectory_builder_3d.cc" not in files
            assert "local_trajectory_builder_3d.h" not in files
    if not os.path.exists("/dev/kfd"):
        run = subprocess.run([LTB3_MI355X], capture_output=True, text=True, timeout=120)
        assert run.returncode != 0 and "no CPU fallback" in run.stderr


LTB3_RESIDENT = os.path.join(DROPIN, "_build", "local_trajectory_builder_3d_resident_mi355x")


@pytest.mark.gpu
@pytest.mark.parametrize("binary", ["adapters", "resident"])
def test_local_trajectory_builder_3d_on_the_gpu_follows_the_references_drive(binary):
    """The same 60 sweeps with the device under the unmodified builder ("resident": both hybrid
    grids of every submap in HBM too, inserted into by cmx_grid3d_insert): three voxel filters
    (bit-exact), the 3D real-time correlative matcher (bit-exact), the 3D Ceres matcher on both
    hybrid grids (the device's solver).  On the builder's boxes the output has been byte-identical
    to the reference-linked build's; what is REQUIRED is the closed-loop bound of the 2D test:
    the first ten poses to 1e-6, all of them to 5 mm, the same insertions, the same submaps to 1 %."""
    path = LTB3_MI355X if binary == "adapters" else LTB3_RESIDENT
    assert os.path.exists(path), "examples/dropin/_build is prebuilt by __graft_entry__.build()"
    out = subprocess.run([path], check=True, capture_output=True, text=True, timeout=600).stdout
    got, got_submaps, got_worst = _drive_3d(out)
    want, want_submaps, want_worst = _drive_3d(open(LTB3_GOLDEN).read())
    assert len(got) == len(want) == 60
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[0] == w[0] and g[4] == w[4]
        np.testing.assert_allclose(g[1], w[1], rtol=0, atol=1e-6 if k < 10 else 5e-3)
        assert abs(g[2] - w[2]) <= 5 and abs(g[3] - w[3]) <= 5
    assert abs(got_worst - want_worst) < 5e-3
    assert len(got_submaps) == len(want_submaps) == 2
    for g, w in zip(got_submaps, want_submaps):
        assert g[:2] == w[:2]
        np.testing.assert_allclose(g[2:], w[2:], rtol=0.01)
    print("identical to the reference-linked build's output" if out == open(LTB3_GOLDEN).read()
          else "within the closed-loop bound of the reference-linked build's output")
