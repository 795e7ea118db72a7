"""C4 (RT-3D): the staged second candidate round against the unstaged one -- same score and
pose, pass timings (debug switch rt3d_report) and wall time.   python tools/stage_probe.py [reps]"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cartographer_amd import _lib, scan_matching_3d as sm3, synth  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
grid, world = synth.make_submap_3d(42, 0.1, (15.0, 15.0, 7.5), 8, 32, 512)
vox = grid.voxels()
pos = world.free_position(77, 0.5)
cloud = world.scan(pos, 0.3, 64, 1024, seed=9)
c, s = math.cos(0.31 / 2), math.sin(0.31 / 2)
init = sm3.Rigid3d(tuple(pos + np.array([0.07, -0.04, 0.02])), (c, 0.0, 0.0, s))
m = sm3.RealTimeCorrelativeScanMatcher3D(0.5, math.radians(2.0), 0.1, 0.1)
results = {}
for staged, verify, report in (("0", "0", "1"), ("1", "0", "1"), ("1", "1", "0"), ("0", "0", "0"),
                               ("1", "0", "0")):
    _lib.debug_set(rt3d_unstaged=1 - int(staged), rt3d_verify=int(verify), rt3d_report=int(report))
    best = 1e9
    for rep in range(reps):
        t0 = time.perf_counter()
        score, est = m.match(init, cloud, 0.1, vox)
        best = min(best, time.perf_counter() - t0)
    key = (np.float32(score), tuple(est.translation), tuple(est.rotation))
    results[(staged, verify, report)] = key
    print(f"staged {staged} verify {verify} report {report}: best wall {best * 1e3:.2f} ms, device "
          f"{m.last_stats['device_ms']:.2f} ms, score {score:.7f}, bounds + candidates "
          f"{m.last_stats['coarse_candidates']}, finalists {m.last_stats['nodes_expanded']}",
          flush=True)
ref = results[("0", "0", "1")]
bad = [k for k, v in results.items() if v != ref]
print("STAGE PROBE", "MISMATCH " + str(bad) if bad else "OK: every mode returns the same score and pose")
sys.exit(1 if bad else 0)
