"""C1 (real-time 2D), the DEVICE side of a large batch on bench.py's distinct inputs: the call as
ONE part (all matches in one launch), HIP-event brackets on, then the in-kernel timeline of the
bound kernel's phases.
   python tools/c1_device_probe.py [batch] [--set name=value ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("batch", nargs="?", type=int, default=1024)
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--no-timeline", action="store_true")
ap.add_argument("--parts", type=int, nargs="*", default=[0, 1, 2])
cli = ap.parse_args()
base = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set}
args = argparse.Namespace(beams=1000, matches=128, c1_distinct=0)
w = bench.Rt2DWorkload(args, 0, matches=cli.batch, grid=200, dirty=False)
for parts in cli.parts:
    _lib.debug_set(rt2d_parts=parts, **base)
    for _ in range(5):
        w.search()
    for timing in (0, 1):
        _lib.debug_set(timing=timing)
        t0 = time.perf_counter()
        dev = ker = 0.0
        reps = 20
        for _ in range(reps):
            r = w.search()
            dev += r[3]["device_ms"]
            ker += r[3]["dominant_kernel_ms"]
        wall = (time.perf_counter() - t0) / reps
        print(f"[{cli.set} parts={parts} timing={timing}] batch {cli.batch}: wall {wall * 1e6:.1f} us, "
              f"device {dev / reps * 1e3:.1f} us, kernel(s) {ker / reps * 1e3:.1f} us", flush=True)
    _lib.debug_set(timing=0)
if not cli.no_timeline:
    _lib.debug_set(rt2d_parts=1, timeline=1, **base)
    for _ in range(2):
        sys.stderr.write("--- timeline, one part\n")
        w.search()
    _lib.debug_set(timeline=0)
sys.stdout.flush()
sys.exit(0)
