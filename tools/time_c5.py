"""C5 (fast-3D) batch timing probe: where a step's wall time goes (python marshalling, the C entry
point, the device).   python tools/time_c5.py [pairs ...]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(submaps=1, beams=1000)
for pairs in [int(a) for a in (sys.argv[1:] or ["32"])]:
    w = bench.Fast3DWorkload(args, 0, pairs=pairs)
    for _ in range(3):
        w.search()
    t0 = time.perf_counter()
    for _ in range(5):
        w.search()
    print(f"pairs {pairs}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per step", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        w.search()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
