"""C5 (fast-3D): a single pair and the 32-pair share, wall / device time per step.
   python tools/c5_probe.py [--set name=value ...] [pairs ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("pairs", nargs="*", type=int, default=[1, 32])
ap.add_argument("--set", action="append", default=[], help="debug switch name=value")
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"
args = argparse.Namespace(submaps=1, beams=1000)
for pairs in cli.pairs:
    w = bench.Fast3DWorkload(args, 0, pairs=pairs)
    for _ in range(5):
        w.search()
    reps = 100 if pairs == 1 else 10
    best, dev = 1e9, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            r = w.search()
        best = min(best, (time.perf_counter() - t0) / reps)
    dev = r[3]["device_ms"]
    print(f"[{tag}] C5 pairs {pairs}: wall {best * 1e3:.3f} ms, device {dev:.3f} ms, found "
          f"{int(sum(r[0]))}, nodes {r[3]['nodes_expanded']}", flush=True)
