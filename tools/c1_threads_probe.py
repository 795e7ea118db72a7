"""C1 from several host threads (bench.py's Rt2DPipelinedWorkload: every thread its own argument
arrays, workspaces and streams): matches/s and candidates/s by threads x matches per call.
   python tools/c1_threads_probe.py [--set name=value ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--shapes", default="128x8,256x8,512x4,1024x2,1024x4,341x8")
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"
args = argparse.Namespace(beams=1000, matches=128, c1_distinct=0)
for shape in cli.shapes.split(","):
    matches, threads = (int(v) for v in shape.split("x"))
    w = bench.Rt2DPipelinedWorkload(args, 0, matches, threads, 4)
    for _ in range(3):
        w.search()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        r = w.search()
    dt = (time.perf_counter() - t0) / reps
    total = matches * threads * 4
    print(f"[{tag}] {threads} threads x 4 calls x {matches} matches ({w.distinct} distinct): "
          f"{dt * 1e6:.0f} us per step, {total / dt:.3e} matches/s, "
          f"{r[3]['candidates_scored'] / dt:.3e} cand/s", flush=True)
    del w
sys.stdout.flush()
os._exit(0)
