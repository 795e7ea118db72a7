"""bench.py's control flow and JSON contract, on CPU: the device calls are replaced by stubs, so
this checks what the driver depends on -- exactly `warmup + steps` matches are issued, ONE JSON
line comes out last, it carries every required key plus `roofline`, and the optional paths
(`--concurrency`, the distributed branch on a 1-rank gloo group) keep the same accounting.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


@pytest.fixture()
def stubbed_bench(monkeypatch):
    import torch
    import torch.distributed as dist
    from cartographer_amd import scan_matching as sm
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    zeros, tensor = torch.zeros, torch.tensor
    strip = lambda k: {key: v for key, v in k.items() if key != "device"}     # noqa: E731
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: zeros(*a, **strip(k)))
    monkeypatch.setattr(torch, "tensor", lambda *a, **k: tensor(*a, **strip(k)))
    init = dist.init_process_group
    monkeypatch.setattr(dist, "init_process_group",
                        lambda backend, device_id=None: init("gloo", rank=0, world_size=1))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29541")
    calls = []

    class FakeMatcher:
        def __init__(self, grid, depth, device=0):
            pass

    def fake_batch(matchers, cloud, min_score):
        calls.append(len(matchers))
        n = len(matchers)
        return (np.ones(n, np.int32), np.full(n, 0.7, np.float32), np.zeros((n, 3)),
                dict(candidates_scored=1000 * n, coarse_candidates=900 * n,
                     dominant_kernel_ms=0.03, device_ms=0.17, num_scans=50 * n,
                     nodes_expanded=30))
    monkeypatch.setattr(sm, "FastCorrelativeScanMatcher2D", FakeMatcher)
    monkeypatch.setattr(sm, "PointCloudOnDevice", lambda scan, device=0: scan)
    monkeypatch.setattr(sm, "match_full_submap_batch", fake_batch)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    import bench

    def run(*argv):
        calls.clear()
        monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--grid", "120"]
                            + list(argv))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        lines = [l for l in buf.getvalue().splitlines() if l.strip()]
        return json.loads(lines[-1]), calls[:], len(lines)
    return run


@pytest.mark.parametrize("extra", [(), ("--concurrency", "3"), ("--force-dist",),
                                   ("--force-dist", "--concurrency", "3"), ("--submaps", "4")])
def test_bench_issues_exactly_the_requested_steps(stubbed_bench, extra):
    out, calls, num_lines = stubbed_bench("--steps", "7", "--warmup", "3", *extra)
    submaps = 4 if "--submaps" in extra else 1
    assert num_lines == 1                                    # one JSON line, nothing after it
    assert calls == [submaps] * 10                           # 3 untimed + exactly 7 timed
    for key in REQUIRED:
        assert key in out, key
    assert out["steps"] == 7 and out["warmup"] == 3 and out["n_gpus"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert out["config"]["candidates_per_step"] == 1000.0 * submaps
    assert out["config"]["host_threads"] == (3 if "--concurrency" in extra else 1)
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 7 - 7000.0 * submaps) < 1e-6
    assert "cpu_baseline" not in out                         # --no-cpu-baseline


def test_bench_cpu_baseline_leg(stubbed_bench, monkeypatch):
    """The `cpu_baseline` object (rank 0, N = 1 only): the reference's own matcher source where
    oracle/_ref is built, otherwise the oracle port; a bounded sample."""
    import bench
    argv = ["bench.py", "--grid", "120", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.3"]
    monkeypatch.setattr(sys, "argv", argv)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    out = json.loads(buf.getvalue().strip().splitlines()[-1])
    base = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in base, key
    assert base["kind"] in ("reference", "port") and base["unit"] == "candidates/s"
    assert base["value"] > 0 and base["cores"] >= 1
