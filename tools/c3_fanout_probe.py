"""One scan against S submaps: the batch call against S single searches issued from T host threads.
   python tools/c3_fanout_probe.py [--submaps 16] [--threads 16]"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import numpy as np  # noqa: E402
from cartographer_amd import _lib, scan_matching as sm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--submaps", type=int, default=16)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--set", action="append", default=[])
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
args = argparse.Namespace(submaps=cli.submaps, grid=400, depth=7, beams=1000, min_score=0.6, scans=8,
                          parity_submaps=1)
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=True)
for k in range(8):
    w.search(k)
t0 = time.perf_counter()
for k in range(cli.reps):
    batch = w.search(k)
dt_batch = (time.perf_counter() - t0) / cli.reps
pool = ThreadPoolExecutor(cli.threads)


def fan(k):
    cloud = w.clouds[k % len(w.clouds)]
    def one(m):
        return sm.match_full_submap_batch([m], cloud, 0.6)
    return list(pool.map(one, w.matchers))


for k in range(8):
    fan(k)
t0 = time.perf_counter()
for k in range(cli.reps):
    out = fan(k)
dt_fan = (time.perf_counter() - t0) / cli.reps
k = cli.reps - 1
found = np.array([int(r[0][0]) for r in out])
scores = np.array([r[1][0] for r in out], np.float32)
assert (found == batch[0]).all() and (scores[found != 0] == batch[1][found != 0]).all()
_lib.debug_set(fast2d_fanout=4)
for k in range(8):
    w.search(k)
t0 = time.perf_counter()
for k in range(cli.reps):
    native = w.search(k)
dt_native = (time.perf_counter() - t0) / cli.reps
_lib.debug_set(fast2d_fanout=0)
assert (native[0] == batch[0]).all() and (native[1][batch[0] != 0] == batch[1][batch[0] != 0]).all()
assert (np.asarray(native[2])[batch[0] != 0] == np.asarray(batch[2])[batch[0] != 0]).all()
print(f"C3 {cli.submaps} submaps: native fan-out over the host pool {dt_native * 1e3:.3f} ms")
print(f"C3 {cli.submaps} submaps: batch call {dt_batch * 1e3:.3f} ms, {cli.threads} threads x single searches "
      f"{dt_fan * 1e3:.3f} ms; found {int(found.sum())} of {len(found)} (results equal)", flush=True)
