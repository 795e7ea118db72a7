"""C5: what a perfect seed would be worth.  The 32-submap share once as the bench runs it, then
again with every pair's min_score set just below the score that pair's search found: the second
run expands only the nodes whose bound reaches the final best, the floor of any exact search on
these bounds.   python tools/c5_seed_ceiling.py [pairs]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cartographer_amd import scan_matching_3d as sm3  # noqa: E402
from cartographer_amd import _lib as _cmx_lib  # noqa: E402
_cmx_lib.debug_set(timing=1)   # cmx_match_stats *_ms are recorded only on request

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = argparse.Namespace(submaps=pairs, beams=1000)
w = bench.Fast3DWorkload(args, 0, pairs=pairs)


def run(mins, reps=4):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        results, stats = sm3.fast3d_match_batch(w.matchers, [w.node] * pairs, [sm3.Rigid3d()] * pairs,
                                                [0] * pairs, list(mins), w.data)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return results, stats, best


res, st, t = run([0.2] * pairs)
scores = [r["score"] if r else None for r in res]
print(f"as shipped : {t * 1e3:7.3f} ms, nodes expanded {st['nodes_expanded']}, device {st['device_ms']:.3f} ms, "
      f"expansion {st['expansion_ms']:.3f} ms, found {sum(s is not None for s in scores)}")
print("scores     :", " ".join("  -- " if s is None else f"{s:.3f}" for s in scores))
mins = [0.2 if s is None else float(np.nextafter(np.float32(s), np.float32(0))) for s in scores]
res2, st2, t2 = run(mins)
same = all((a is None) == (b is None) and (a is None or a["score"] == b["score"]) for a, b in zip(res, res2))
print(f"perfect seed: {t2 * 1e3:7.3f} ms, nodes expanded {st2['nodes_expanded']}, device {st2['device_ms']:.3f} ms, "
      f"expansion {st2['expansion_ms']:.3f} ms, same results {same}")
# and per pair: which searches carry the nodes
per = []
for i in range(pairs):
    r, s = sm3.fast3d_match_batch([w.matchers[i]] * 2, [w.node] * 2, [sm3.Rigid3d()] * 2, [0, 0],
                                  [0.2, mins[i]], w.data)
    a = w.matchers[i].match(w.node, sm3.Rigid3d(), w.data, 0.2)
    na = w.matchers[i].last_stats["nodes_expanded"]
    b = w.matchers[i].match(w.node, sm3.Rigid3d(), w.data, mins[i])
    nb = w.matchers[i].last_stats["nodes_expanded"]
    per.append((i, scores[i], na, nb))
print("pair score  nodes(as shipped)  nodes(perfect seed)")
for i, s, na, nb in per:
    print(f"{i:4d} {'  -- ' if s is None else f'{s:.3f}'} {na:10d} {nb:10d}")
