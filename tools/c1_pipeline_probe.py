"""C1: 128-match resident batches from one host thread and from several (bench.py's two C1
batch legs, without the rest of the bench).   python tools/c1_pipeline_probe.py [threads ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request

args = argparse.Namespace(matches=128, beams=1000)
single = bench.Rt2DWorkload(args, 0, matches=128)
for _ in range(20):
    single.search()
t0 = time.perf_counter()
for _ in range(200):
    stats = single.search()[3]
dt = (time.perf_counter() - t0) / 200
cand = stats["candidates_scored"]
print(f"one thread: {dt * 1e6:.1f} us per 128-match call, {cand / dt:.3e} candidates/s, device "
      f"{stats['device_ms'] * 1e3:.1f} us, bulk kernel {stats['dominant_kernel_ms'] * 1e3:.1f} us",
      flush=True)
for threads in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    w = bench.Rt2DPipelinedWorkload(args, 0, 128, threads, 8)
    for _ in range(3):
        w.search()
    t0 = time.perf_counter()
    for _ in range(10):
        stats = w.search()[3]
    dt = (time.perf_counter() - t0) / 10
    calls = threads * 8
    print(f"{threads} threads: {dt / calls * 1e6:.1f} us per 128-match call, "
          f"{stats['candidates_scored'] / dt:.3e} candidates/s", flush=True)
