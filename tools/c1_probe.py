"""C1 (real-time 2D) probe on bench.py's workload -- DISTINCT (grid, scan, pose) triples, 1000-point
scans: wall / device / bulk-kernel time per call for each batch size.
   python tools/c1_probe.py [batch ...] [--grid 200] [--dirty] [--distinct N] [--timing]
                            [--set name=value ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("batches", nargs="*", type=int, default=[1, 128, 1024])
ap.add_argument("--grid", type=int, default=200)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--dirty", action="store_true", help="bump every grid's version before each call")
ap.add_argument("--distinct", type=int, default=0)
ap.add_argument("--timing", action="store_true", help="HIP-event brackets on (device / kernel ms)")
ap.add_argument("--set", action="append", default=[], help="debug switch name=value")
cli = ap.parse_args()
if cli.set:
    _lib.debug_set(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in cli.set})
tag = " ".join(cli.set) or "default"
args = argparse.Namespace(beams=1000, matches=128, c1_distinct=cli.distinct)
for batch in cli.batches:
    w = bench.Rt2DWorkload(args, 0, matches=batch, grid=cli.grid, dirty=cli.dirty)
    for _ in range(5):
        w.search()
    _lib.debug_set(timing=1 if cli.timing else 0)
    w.insert_s = 0.0
    t0 = time.perf_counter()
    dev = ker = 0.0
    for _ in range(cli.reps):
        r = w.search()
        dev += r[3]["device_ms"]
        ker += r[3]["dominant_kernel_ms"]
    wall = (time.perf_counter() - t0 - w.insert_s) / cli.reps
    _lib.debug_set(timing=0)
    st = r[3]
    cand = st["candidates_scored"]
    print(f"[{tag}] C1 grid {cli.grid} batch {batch:5d} ({w.distinct} distinct, {w.n_points} pts)"
          f"{' dirty' if cli.dirty else ''}: wall {wall * 1e6:8.1f} us, device "
          f"{dev / cli.reps * 1e3:7.1f} us, bulk kernel {ker / cli.reps * 1e3:7.1f} us -> "
          f"{cand / wall:.3e} cand/s; per match: summed {st.get('coarse_candidates', 0) / batch:.0f} "
          f"refined {st.get('refined_candidates', 0) / batch:.1f}, f32 finalists "
          f"{st.get('finalists', 0) / batch:.1f}; score[0] {r[1][0]:.6f}", flush=True)
    del w
sys.stdout.flush()
sys.exit(0)
