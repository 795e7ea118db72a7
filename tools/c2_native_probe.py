"""C2 over 8 scans: T Python threads vs T native threads (csrc/host/thread_driver.cc).
   python tools/c2_native_probe.py [T ...] [--set name=value ...]"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("threads", nargs="*", type=int, default=[8, 12, 16])
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--scans", type=int, default=8)
ap.add_argument("--grid", type=int, default=400)
ap.add_argument("--beams", type=int, default=1000)
ap.add_argument("--depth", type=int, default=7)
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"
args = argparse.Namespace(submaps=0, grid=cli.grid, depth=cli.depth, beams=cli.beams, min_score=0.6, scans=cli.scans,
                          parity_submaps=1)
w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=False)
for k in range(4 * cli.scans):
    w.search(k)
per = 400
for T in cli.threads:
    pool = ThreadPoolExecutor(T)

    def worker(t):
        for j in range(per):
            w.search(t * per + j)
    list(pool.map(worker, range(T)))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        list(pool.map(worker, range(T)))
        best = min(best, (time.perf_counter() - t0) / (T * per))
    synth.threaded_full_submap_searches(w.matchers, w.clouds, 0.6, T, 50)
    nbest = 1e9
    for _ in range(3):
        secs, cand_n, found_n = synth.threaded_full_submap_searches(w.matchers, w.clouds, 0.6, T, per)
        nbest = min(nbest, secs / (T * per))
    print(f"[{tag}] C2 {cli.scans} scans x {T} threads: python {best * 1e6:.1f} us per search "
          f"({1 / best:.0f}/s), native {nbest * 1e6:.1f} us ({1 / nbest:.0f}/s, "
          f"{cand_n / (T * per) / nbest:.3e} cand/s), found {found_n} of {T * per}", flush=True)
    pool.shutdown()
