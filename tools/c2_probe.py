"""C2 single stream and the 16-submap share of C3: wall / device time per search.
   python tools/c2_probe.py"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = argparse.Namespace(submaps=0, grid=400, depth=7, beams=1000, min_score=0.6, scans=1)
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=False)
for _ in range(30):
    w.search()
t0 = time.perf_counter()
dev = 0.0
for _ in range(300):
    r = w.search()
    dev += r[3]["device_ms"]
dt = (time.perf_counter() - t0) / 300
print(f"C2 single: wall {dt * 1e6:.1f} us, device {dev / 300 * 1e3:.1f} us, found {int(r[0][0])}, "
      f"score {float(r[1][0]):.7f}, candidates {r[3]['candidates_scored']}, nodes "
      f"{r[3]['nodes_expanded']}", flush=True)
args.submaps = 16
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=True)
for _ in range(3):
    w.search()
t0 = time.perf_counter()
for _ in range(10):
    r = w.search()
dt = (time.perf_counter() - t0) / 10
print(f"C3 16-share: wall {dt * 1e3:.3f} ms, device {r[3]['device_ms']:.3f} ms, expansion "
      f"{r[3]['expansion_ms']:.3f} ms, found {int(sum(r[0]))} of {len(r[0])}", flush=True)
