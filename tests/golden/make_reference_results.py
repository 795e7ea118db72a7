#!/usr/bin/env python3
"""Generates tests/golden/reference_results.json: what the REFERENCE'S OWN scan-matcher sources
(compiled unmodified by `make -C oracle ref`, oracle/ref_shims/README.md) return on the workloads
of tests/golden/workloads.py.  Needs /root/reference or a prebuilt oracle/_ref/libref.so; the
results are committed so that the device can be compared with them on the GPU box.
    python tests/golden/make_reference_results.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def build():
    import workloads as w
    from cartographer_amd import synth
    from oracle import pyoracle as orc
    if orc.ref_lib() is None:
        raise SystemExit("oracle/_ref is not built and /root/reference is absent")
    out = {}

    b = w.fast2d_bench(synth)
    lim = b["lim"]
    m = orc.ReferenceFastCorrelativeScanMatcher2D(b["cells"], lim["resolution"], lim["max_x"],
                                                  lim["max_y"], b["depth"])
    r = m.match_full_submap(b["scan"], 0.6)
    out["fast2d_full_submap"] = dict(found=r["found"], score=r["score"], pose=list(r["pose"]))
    for name, min_score in (("fast2d_windowed", 0.55), ("fast2d_windowed_unreachable", 0.99)):
        r = m.match(b["init"], b["scan"], min_score)
        out[name] = dict(found=r["found"], score=r.get("score"),
                         pose=list(r["pose"]) if r["found"] else None)

    c = w.rt2d_c1(synth)
    lim = c["lim"]
    r = orc.ref_rt2d_match(c["cells"], lim["resolution"], lim["max_x"], lim["max_y"], c["init"],
                           c["scan"], c["lin"], c["ang"], c["tw"], c["rw"])
    out["rt2d_c1"] = dict(score=r["score"], pose=list(r["pose"]))

    t = w.rt2d_tsdf()
    r = orc.ref_rt2d_match(t["tsd"], t["res"], t["max_x"], t["max_y"], t["init"], t["cloud"],
                           t["lin"], t["ang"], t["tw"], t["rw"], weight_cells=t["weight"],
                           truncation_distance=t["truncation"], max_weight=t["max_weight"])
    out["rt2d_tsdf"] = dict(score=r["score"], pose=list(r["pose"]))

    d = w.rt3d(synth)
    r = orc.ref_rt3d_match(d["res"], d["vox"], d["init"], d["cloud"], d["lin"], d["ang"],
                           d["tw"], d["rw"])
    out["rt3d"] = dict(score=r["score"], pose=list(r["pose"]))

    f = w.fast3d(synth)
    o = f["options"]
    m3 = orc.ReferenceFastCorrelativeScanMatcher3D(f["res"], f["vox"], f["low_res"], f["low_vox"],
                                                   f["hist"], o["depth"], o["frd"], o["min_rot"],
                                                   o["min_low"], o["lin_xy"], o["lin_z"], o["ang"])
    r = m3.match(f["node_pose"], f["submap_pose"], f["gravity"], f["hi"], f["lo"],
                 f["scan_hist"], f["min_score"])
    out["fast3d"] = dict(found=r["found"], score=r["score"], pose=list(r["pose"]),
                         rotational_score=r["rotational_score"],
                         low_resolution_score=r["low_resolution_score"])
    return json.loads(json.dumps(out, default=float))


if __name__ == "__main__":
    path = os.path.join(HERE, "reference_results.json")
    with open(path, "w") as f:
        json.dump(build(), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)
