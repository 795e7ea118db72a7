#!/usr/bin/env python3
"""In-kernel timelines (debug switch timeline) + stage traces (trace) of one C2 match, one C1
match and one C1 batch: where the time of the front-end kernels goes.  Prints to stderr.
Usage: python tools/timeline_probe.py [c2] [c1] [c1b]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cartographer_amd import _lib, grid_2d, scan_matching as sm, synth  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request

_lib.debug_set(timeline=1, trace=1)

which = sys.argv[1:] or ["c2", "c1", "c1b"]
if "c2" in which:
    cells, lim, world = synth.make_submap(42, 400, 400, 0.05, 30, 1000, 30.0, 0.01)
    pose = world.free_pose(1234, 0.5)
    scan = world.scan(pose, 1000, 30.0, 0.01, 7)
    m = sm.FastCorrelativeScanMatcher2D(sm.Grid2D(cells, 0.05, lim["max_x"], lim["max_y"]), 7)
    cloud = sm.PointCloudOnDevice(scan)
    for _ in range(4):
        print("--- C2 match", file=sys.stderr)
        t0 = time.perf_counter()
        sm.match_full_submap_batch([m], cloud, 0.6)
        print(f"    wall {1e6 * (time.perf_counter() - t0):.1f} us", file=sys.stderr)
if "c1" in which or "c1b" in which:
    rt = sm.RealTimeCorrelativeScanMatcher2D(0.3, math.radians(7.0), 0.1, 0.1)
    G, I, S = [], [], []
    for k in range(8):
        cells, lim, world = synth.make_submap(42 + k, 200, 200, 0.05, 30, 1000, 5.0, 0.01)
        pose = world.free_pose(1234, 0.5)
        G.append(grid_2d.ProbabilityGridOnDevice(0.05, (lim["max_x"], lim["max_y"]), 200, 200,
                                                 cells=cells))
        S.append(world.scan(pose, 1000, 5.0, 0.01, 7))
        I.append(sm.Rigid2d(pose[0] + 0.12, pose[1] - 0.08, pose[2] + math.radians(3.0)))
    if "c1" in which:
        for _ in range(4):
            print("--- C1 single", file=sys.stderr)
            t0 = time.perf_counter()
            rt.match(I[0], S[0], G[0])
            print(f"    wall {1e6 * (time.perf_counter() - t0):.1f} us; {rt.last_stats}",
                  file=sys.stderr)
    if "c1b" in which:
        for batch in (32, 128, 512):
            g = [G[i % 8] for i in range(batch)]
            i_ = [I[i % 8] for i in range(batch)]
            s_ = [S[i % 8] for i in range(batch)]
            for rep in range(3):
                print(f"--- C1 batch {batch}", file=sys.stderr)
                t0 = time.perf_counter()
                _, _, st = sm.rt2d_match_batch(rt, g, i_, s_)
                print(f"    wall {1e6 * (time.perf_counter() - t0):.1f} us; kernel "
                      f"{st['dominant_kernel_ms'] * 1e3:.1f} us device {st['device_ms'] * 1e3:.1f} us -> "
                      f"{st['candidates_scored'] / (st['dominant_kernel_ms'] * 1e-3):.3e} cand/s kernel",
                      file=sys.stderr)
