"""C2 over several DIFFERENT scans (poses of the same world): per scan, single stream -- wall,
device time, nodes, candidates, and (--trace) the stage durations and list sizes of one search;
then all scans from T host threads (the bench headline's shape).
   python tools/c2m_probe.py [--scans 8] [--threads 8] [--trace] [--set name=value ...]"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--set", action="append", default=[], help="debug switch name=value")
ap.add_argument("--scans", type=int, default=8)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--trace", action="store_true")
ap.add_argument("--reps", type=int, default=200)
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"

args = argparse.Namespace(submaps=0, grid=400, depth=7, beams=1000, min_score=0.6, scans=cli.scans,
                          parity_submaps=1)
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=False)
for k in range(cli.scans):
    for _ in range(10):
        w.search(k)
_lib.debug_set(timing=1)
total_wall = 0.0
for k in range(cli.scans):
    t0 = time.perf_counter()
    dev = 0.0
    for _ in range(cli.reps):
        r = w.search(k)
        dev += r[3]["device_ms"]
    dt = (time.perf_counter() - t0) / cli.reps
    total_wall += dt
    print(f"[{tag}] scan {k}: wall {dt * 1e6:.1f} us, device {dev / cli.reps * 1e3:.1f} us, found "
          f"{int(r[0][0])}, score {float(r[1][0]):.6f}, candidates {r[3]['candidates_scored']}, "
          f"coarse {r[3]['coarse_candidates']}, nodes {r[3]['nodes_expanded']}", flush=True)
    if cli.trace:
        _lib.debug_set(trace=1)
        w.search(k)
        _lib.debug_set(trace=0)
print(f"[{tag}] mean single-stream wall {total_wall / cli.scans * 1e6:.1f} us", flush=True)
_lib.debug_set(timing=0)
if cli.threads > 1:
    T, per = cli.threads, 200
    pool = ThreadPoolExecutor(T)
    cands = [0] * T

    def worker(t):
        c = 0
        for j in range(per):
            c += w.search(t + j)[3]["candidates_scored"]
        cands[t] = c
    list(pool.map(worker, range(T)))
    best, cand = 1e9, 0
    for _ in range(3):
        t0 = time.perf_counter()
        list(pool.map(worker, range(T)))
        dt = time.perf_counter() - t0
        if dt < best:
            best, cand = dt, sum(cands)
    print(f"[{tag}] {cli.scans} scans x {T} threads: {best / (T * per) * 1e6:.1f} us per search, "
          f"{T * per / best:.0f} matches/s, {cand / best:.3e} candidates/s", flush=True)
sys.stdout.flush()
os._exit(0)     # (skip the interpreter's teardown of ctypes-owned handles: noise on stderr)
