#!/usr/bin/env python3
"""Generates tests/golden/rt3d_c4_reference.json: what the REFERENCE'S OWN
real_time_correlative_scan_matcher_3d.cc (compiled unmodified into oracle/_ref) returns on

  rt3d_c4         BASELINE config[3] at its own window -- 65 536 points, 150^3 HybridGrid,
                  +-0.5 m / +-2 deg: 1 771 561 candidates, 1.16e11 transformed points.  The
                  reference's GenerateExhaustiveSearchTransforms / TransformPointCloud /
                  ScoreCandidate run over candidate ranges on host threads
                  (oracle/ref_wrapper_rt3d_mt.cc; ~10 minutes on 8 cores), joined by Match's own
                  first-maximum rule;
  rt3d_c4_shaped  the same SHAPE (L = 5, 216 groups of 2x2x2 translations per rotation, A = 3,
                  tilted initial orientation) at 4096 points: here the unthreaded original
                  (`RealTimeCorrelativeScanMatcher3D::Match` itself) is run too and must agree
                  with the threaded ranges bit for bit.

    python tests/golden/make_rt3d_c4_golden.py [--threads 8] [--full-match] [--skip-c4]
`--full-match` also runs the unthreaded Match on C4 itself (more than an hour on one core).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
PATH = os.path.join(HERE, "rt3d_c4_reference.json")


def run(d, threads, full_match):
    from oracle import pyoracle as orc
    t = time.time()
    r = orc.ref_rt3d_match_mt(d["res"], d["vox"], d["init"], d["cloud"], d["lin"], d["ang"],
                              d["tw"], d["rw"], threads)
    out = dict(score=r["score"], pose=list(r["pose"]), best_index=r["best_index"],
               num_candidates=r["num_candidates"], num_points=int(len(d["cloud"])),
               generated_by=f"ref_rt3d_match_mt, {threads} threads, {time.time() - t:.0f} s")
    if full_match:
        t = time.time()
        m = orc.ref_rt3d_match(d["res"], d["vox"], d["init"], d["cloud"], d["lin"], d["ang"],
                               d["tw"], d["rw"])
        assert np.float32(m["score"]) == np.float32(r["score"]), (m, r)
        assert np.array_equal(m["pose"], r["pose"]), (m, r)
        out["unthreaded_match_agrees"] = True
        out["generated_by"] += f"; RealTimeCorrelativeScanMatcher3D::Match itself, {time.time() - t:.0f} s: identical"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--full-match", action="store_true")
    ap.add_argument("--skip-c4", action="store_true")
    args = ap.parse_args()
    import workloads as w
    from cartographer_amd import synth
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        raise SystemExit("oracle/_ref is not built and /root/reference is absent")
    out = {}
    if os.path.exists(PATH):
        out = json.load(open(PATH))
    out["rt3d_c4_shaped"] = run(w.rt3d_c4_shaped(synth), args.threads, True)
    print("shaped:", out["rt3d_c4_shaped"], flush=True)
    if not args.skip_c4:
        out["rt3d_c4"] = run(w.rt3d_c4(synth), args.threads, args.full_match)
        print("c4:", out["rt3d_c4"], flush=True)
    with open(PATH, "w") as f:
        json.dump(json.loads(json.dumps(out, default=float)), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", PATH)


if __name__ == "__main__":
    main()
