"""C2 single stream, C2 from T host threads, and the 16-submap share of C3: wall / device time
per search.
   python tools/c2_probe.py [--set name=value ...] [--threads 8] [--no-c3]"""
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
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--no-c3", action="store_true")
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"

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
print(f"[{tag}] C2 single: wall {dt * 1e6:.1f} us, device {dev / 300 * 1e3:.1f} us, found "
      f"{int(r[0][0])}, score {float(r[1][0]):.7f}, candidates {r[3]['candidates_scored']}, nodes "
      f"{r[3]['nodes_expanded']}", flush=True)
if cli.threads > 1:
    T, per = cli.threads, 400
    pool = ThreadPoolExecutor(T)

    def worker(_):
        for _ in range(per):
            w.search()
    list(pool.map(worker, range(T)))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        list(pool.map(worker, range(T)))
        best = min(best, (time.perf_counter() - t0) / (T * per))
    cand = r[3]["candidates_scored"]
    print(f"[{tag}] C2 x {T} threads: {best * 1e6:.1f} us per search -> {cand / best:.3e} "
          f"candidates/s", flush=True)
    # the same from native threads (csrc/host/thread_driver.cc): no interpreter lock between calls
    from cartographer_amd import synth
    synth.threaded_full_submap_searches(w.matchers, w.clouds, 0.6, T, 50)
    best = 1e9
    for _ in range(3):
        secs, cand_n, found_n = synth.threaded_full_submap_searches(w.matchers, w.clouds, 0.6, T, per)
        best = min(best, secs / (T * per))
    print(f"[{tag}] C2 x {T} native threads: {best * 1e6:.1f} us per search -> "
          f"{cand_n / (T * per) / best:.3e} candidates/s, found {found_n} of {T * per}", flush=True)
if not cli.no_c3:
    args.submaps = 16
    w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=True)
    for _ in range(3):
        w.search()
    t0 = time.perf_counter()
    for _ in range(10):
        r = w.search()
    dt = (time.perf_counter() - t0) / 10
    print(f"[{tag}] C3 16-share: wall {dt * 1e3:.3f} ms, device {r[3]['device_ms']:.3f} ms, "
          f"expansion {r[3]['expansion_ms']:.3f} ms, found {int(sum(r[0]))} of {len(r[0])}",
          flush=True)
