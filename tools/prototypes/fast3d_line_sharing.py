"""C5 (fast 3D): how many 128-byte oct lines the branch-and-bound expansion would fetch if
neighbouring nodes were expanded together (order-preserving frontier).

The oracle's search (the reference's depth-first schedule: more nodes than the device's
level-synchronous one, the same kind of frontier) dumps every expanded node and the discrete scans
(ORC_DUMP_NODES); per (scan, depth) the nodes are sorted by offset and grouped in different ways,
and the distinct lines of one group's lookups are counted.  A lookup of child depth d reads the
8-byte oct word at X = cell.x + (offset.x >> e) (+ the same for y, z), 16 words per line along x.

    python tools/prototypes/fast3d_line_sharing.py [dump file]
"""
import argparse
import math
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(tempfile.gettempdir(), "fast3d_nodes.bin")
if not os.path.exists(path):
    args = argparse.Namespace(submaps=1, beams=1000)
    w = bench.Fast3DWorkload.__new__(bench.Fast3DWorkload)
    # (the workload's host-side data only: the constructor would build device matchers)
    from cartographer_amd import synth
    size = (15.0, 15.0, 7.5)
    grid, world = synth.make_submap_3d(42, 0.1, size, 8, 32, 512)
    low, _ = synth.make_submap_3d(42, 0.45, size, 8, 32, 512)
    vox, low_vox = grid.voxels(), low.voxels()
    rng = np.random.default_rng(1)
    hist = rng.uniform(0.0, 1.0, 120).astype(np.float32)
    hist[10:14] += 6.0
    pos = world.free_position(77, 0.6)
    yaw = 0.4
    full = world.scan(pos, yaw, 32, 512, seed=1)
    hi, lo = full[::6].copy(), full[::80].copy()
    scan_hist = np.roll(hist, -19).copy()
    m = orc.FastCorrelativeScanMatcher3D(0.1, vox, 0.45, low_vox, hist, 8, 3, 0.77, 0.35, 5.0, 1.0,
                                         math.radians(15.0))
    node = [pos[0] + 0.8, pos[1] - 0.6, pos[2] + 0.2, math.cos((yaw + 0.1) / 2), 0.0, 0.0,
            math.sin((yaw + 0.1) / 2)]
    os.environ["ORC_DUMP_NODES"] = path
    got = m.match(node, [0, 0, 0, 1, 0, 0, 0], [1, 0, 0, 0], hi, lo, scan_hist, 0.2)
    del os.environ["ORC_DUMP_NODES"]
    print("oracle search:", {k: got[k] for k in ("found", "score", "nodes_expanded", "num_scans")})

raw = np.fromfile(path, np.int32)
num_scans, depth, n, full_depth = raw[:4]
cells = raw[4:4 + num_scans * depth * n * 3].reshape(num_scans, depth, n, 3)
nodes = raw[4 + num_scans * depth * n * 3:].reshape(-1, 5)
print(f"{num_scans} scans, depth {depth}, {n} points, {len(nodes)} expanded nodes")


TILE = (4, 0, 0)       # log2 of the words of one 128-byte line along x, y, z (16 words in all)


def lines_of(scan, child_depth, offsets):
    """Distinct oct lines of the lookups of a group of nodes (offsets [k, 3]) at one child depth."""
    e = max(0, child_depth - full_depth + 1)
    c = cells[scan, child_depth]                                  # [n, 3]
    o = offsets >> e                                              # [k, 3]
    X = c[None, :, 0] + o[:, None, 0]
    Y = c[None, :, 1] + o[:, None, 1]
    Z = c[None, :, 2] + o[:, None, 2]
    key = (((Z.astype(np.int64) + 4096) >> TILE[2]) << 40) | \
          (((Y.astype(np.int64) + 4096) >> TILE[1]) << 20) | ((X.astype(np.int64) + 4096) >> TILE[0])
    return len(np.unique(key)), key.size


for TILE in ((4, 0, 0), (2, 1, 1), (2, 2, 0), (1, 1, 2), (3, 1, 0)):
    print(f"---- line = {1 << TILE[0]} x {1 << TILE[1]} x {1 << TILE[2]} oct words (x, y, z)")
    total = {"lookups": 0, "single": 0, "family": 0, "xrun8": 0, "block": 0}
    per_depth = {}
    for d in range(depth - 1, 0, -1):                                # node depth d expands children d - 1
        at = nodes[nodes[:, 1] == d]
        if not len(at):
            continue
        step = 1 << d                                                 # the lattice of depth-d nodes
        stats = {"nodes": len(at), "lookups": 0, "single": 0, "family": 0, "xrun8": 0, "block": 0}
        for scan in np.unique(at[:, 0]):
            off = at[at[:, 0] == scan][:, 2:5]
            order = np.lexsort((off[:, 0], off[:, 1], off[:, 2]))
            off = off[order]
            # every node on its own
            for k in range(len(off)):
                u, q = lines_of(scan, d - 1, off[k:k + 1])
                stats["single"] += u
                stats["lookups"] += q
            # families: the expanded children of one parent (same 2 x 2 x 2 block of the lattice)
            parent = off // (2 * step)
            _, fam = np.unique(parent, axis=0, return_inverse=True)
            for f in np.unique(fam):
                stats["family"] += lines_of(scan, d - 1, off[fam.ravel() == f])[0]
            # runs of up to eight x-neighbours in one lattice row
            row = np.stack([off[:, 1], off[:, 2], off[:, 0] // (8 * step)], 1)
            _, run = np.unique(row, axis=0, return_inverse=True)
            for r in np.unique(run):
                stats["xrun8"] += lines_of(scan, d - 1, off[run.ravel() == r])[0]
            # blocks of 8 x 2 x 2 lattice cells
            blk = np.stack([off[:, 0] // (8 * step), off[:, 1] // (2 * step), off[:, 2] // (2 * step)], 1)
            _, bl = np.unique(blk, axis=0, return_inverse=True)
            for b in np.unique(bl):
                stats["block"] += lines_of(scan, d - 1, off[bl.ravel() == b])[0]
        per_depth[d] = stats
        for k in total:
            total[k] += stats[k]
        print(f"node depth {d}: {stats['nodes']} nodes, lines per lookup: alone {stats['single'] / stats['lookups']:.3f}, "
              f"families {stats['family'] / stats['lookups']:.3f}, x-runs of 8 {stats['xrun8'] / stats['lookups']:.3f}, "
              f"8x2x2 blocks {stats['block'] / stats['lookups']:.3f}")
    print(f"all depths: lookups {total['lookups']}, lines per lookup: alone {total['single'] / total['lookups']:.3f}, "
          f"families {total['family'] / total['lookups']:.3f}, x-runs of 8 {total['xrun8'] / total['lookups']:.3f}, "
          f"8x2x2 blocks {total['block'] / total['lookups']:.3f}")
