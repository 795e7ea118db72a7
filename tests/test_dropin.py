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
