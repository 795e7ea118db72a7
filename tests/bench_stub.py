"""Runs bench.main() with the device calls replaced by stubs (CPU only): used by
tests/test_bench_contract.py, in-process and as `python tests/bench_stub.py <bench flags>` for the
two-rank gloo run.  The stubs stand in for torch.cuda, the RCCL backend (gloo instead) and the
three scan_matching entry points bench.py calls; everything else is the real script.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CALLS = []


def install(setattr_=setattr):
    import torch
    import torch.distributed as dist
    from cartographer_amd import scan_matching as sm
    os.environ["CMX_BENCH_TORCH_DEVICE"] = "cpu"
    setattr_(torch.cuda, "is_available", lambda: True)
    setattr_(torch.cuda, "set_device", lambda d: None)
    setattr_(torch.cuda, "synchronize", lambda *a, **k: None)
    zeros, tensor = torch.zeros, torch.tensor
    strip = lambda k: {key: v for key, v in k.items() if key != "device"}     # noqa: E731
    setattr_(torch, "zeros", lambda *a, **k: zeros(*a, **strip(k)))
    setattr_(torch, "tensor", lambda *a, **k: tensor(*a, **strip(k)))
    init = dist.init_process_group
    setattr_(dist, "init_process_group", lambda backend, device_id=None: init("gloo"))

    class FakeMatcher:
        def __init__(self, grid, depth, device=0):
            pass

    # CMX_STUB_SLEEP_US: a search takes that long (per rank: unequal ranks must still agree
    # on the calibrated pass count)
    sleep_s = float(os.environ.get("CMX_STUB_SLEEP_US", "0")) * 1e-6

    def fake_batch(matchers, cloud, min_score):
        CALLS.append(len(matchers))
        if sleep_s:
            import time
            time.sleep(sleep_s)
        n = len(matchers)
        return (np.ones(n, np.int32), np.full(n, 0.7, np.float32), np.zeros((n, 3)),
                dict(candidates_scored=1000 * n, coarse_candidates=900 * n,
                     dominant_kernel_ms=0.03, device_ms=0.17, num_scans=50 * n,
                     nodes_expanded=30))
    # C5 (--config c5): the 3D matcher, its batch call and the 150^3 synthetic submaps (a
    # small world stands in: the contract test is about sharding, collectives and the JSON line)
    from cartographer_amd import scan_matching_3d as sm3, synth

    class FakeMatcher3D:
        last_stats = {}

        def __init__(self, *a, **k):
            pass

    def fake_batch_3d(matchers, nodes, submaps, full, thresholds, data):
        CALLS.append(len(matchers))
        n = len(matchers)
        got = [dict(score=0.5 + 0.01 * i, pose_estimate=nodes[i], rotational_score=1.0,
                    low_resolution_score=1.0) if i % 2 == 0 else None for i in range(n)]
        return got, dict(candidates_scored=2000 * n, coarse_candidates=1500 * n,
                         dominant_kernel_ms=0.05, device_ms=0.4, num_scans=7 * n,
                         nodes_expanded=300 * n, expansion_ms=0.3, expansion_launches=7,
                         expansion_nodes=250 * n, expansion_lookups=250 * n * 64)
    real_submap_3d = synth.make_submap_3d
    setattr_(synth, "make_submap_3d",
             lambda seed, res, size, *a: real_submap_3d(seed, max(res, 0.4), (4.0, 4.0, 2.0), 2, 4, 32))
    setattr_(sm3, "FastCorrelativeScanMatcher3D", FakeMatcher3D)
    setattr_(sm3, "fast3d_match_batch", fake_batch_3d)
    setattr_(sm, "FastCorrelativeScanMatcher2D", FakeMatcher)
    setattr_(sm, "PointCloudOnDevice", lambda scan, device=0: scan)
    setattr_(sm, "match_full_submap_batch", fake_batch)
    # The stub's results are made up: the device-vs-reference gate in front of the timed region
    # is replaced by a record (tests/test_bench_contract.py tests the real gate on its own).
    import bench
    setattr_(bench, "parity_gate", lambda workload, result=None: {
        "vs": "stub", "checked": 1, "max_abs_dscore": 0.0, "max_abs_dpose": 0.0,
        "bit_exact": True, "tol": bench.PARITY_TOL})


if __name__ == "__main__":
    install()
    import bench
    sys.argv = ["bench.py"] + sys.argv[1:]
    bench.main()
    print(f"# rank {os.environ.get('RANK', '0')} issued {len(CALLS)} matches", file=sys.stderr)
