"""bench.py's control flow and JSON contract, on CPU: the device calls are replaced by stubs, so
this checks what the driver depends on -- exactly `warmup + steps` matches (+ one untimed instrumented step) are issued, ONE JSON
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
    import bench_stub
    bench_stub.install(monkeypatch.setattr)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29541")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    calls = bench_stub.CALLS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    import bench

    def run(*argv):
        calls.clear()
        monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--no-other", "--grid",
                                          "120", "--passes-per-step", "1", "--concurrency", "1"]
                            + list(argv))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        lines = [l for l in buf.getvalue().splitlines() if l.strip()]
        return json.loads(lines[-1]), calls[:], len(lines)
    return run


@pytest.mark.parametrize("extra", [(), ("--concurrency", "3"), ("--force-dist", "--submaps", "2"),
                                   ("--force-dist", "--concurrency", "3", "--submaps", "3"),
                                   ("--submaps", "4"), ("--config", "c3", "--submaps", "5")])
def test_bench_issues_exactly_the_requested_steps(stubbed_bench, extra):
    out, calls, num_lines = stubbed_bench("--steps", "7", "--warmup", "3", *extra)
    submaps = int(extra[extra.index("--submaps") + 1]) if "--submaps" in extra else 1
    sharded = "--force-dist" in extra or "c3" in extra
    assert num_lines == 1                                    # one JSON line, nothing after it
    # 3 untimed + exactly 7 timed + 1 untimed instrumented step behind the timed region (the
    # library's HIP-event brackets are off inside it: bench.set_timing)
    assert calls == [submaps] * 11
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
    # the dominant kernel is bound by L2 -> L1 gathers, not by HBM: an on-chip peak, frac <= 1
    # on hardware, and the HBM-side ratios next to it
    assert roof["bound"] == "l2-gather" and roof["unit"] == "GB/s" and roof["peak"] == 34500.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5 * roof["frac"] + 1e-12
    for key in ("kernel", "kernel_ms", "algorithmic_bytes", "hbm_frac_algorithmic",
                "hbm_frac_traffic"):
        assert key in roof, key
    assert out["config"]["name"] == ("c3" if sharded else "c2")
    if sharded:      # the reference's semantics (every constraint to every rank) + the best match
        assert out["config"]["constraints_found_node_wide"] == submaps
        assert out["config"]["best_match"]["submap"] == 0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 7 / (7000.0 * submaps) - 1) < 1e-4
    assert "cpu_baseline" not in out                         # --no-cpu-baseline
    # a step is `passes_per_step` passes; the compact per-config summary closes the line
    assert out["config"]["passes_per_step"] == 1 and list(out)[-1] == "summary"
    assert abs(out["summary"][out["config"]["name"]]["cand_per_s"] / out["value"] - 1) < 1e-3
    assert out["parity"]["checked"] == 1 and out["summary"][out["config"]["name"]]["parity"] == "exact"


def test_bench_cpu_baseline_leg(stubbed_bench, monkeypatch):
    """The `cpu_baseline` object (rank 0, N = 1 only): the reference's own matcher source where
    oracle/_ref is built, otherwise the oracle port; a bounded sample."""
    import bench
    argv = ["bench.py", "--grid", "120", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.3",
            "--no-other"]
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


def test_bench_two_ranks_gloo():
    """The N > 1 path as the driver launches it (one process per rank, RANK / WORLD_SIZE /
    MASTER_* from the environment), on two CPU processes over gloo: only rank 0 prints, ONE line,
    `n_gpus` 2, the work of both ranks summed, one all-reduce per step on every rank."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(root, "tests", "bench_stub.py"), "--gpus", "2",
             "--steps", "5", "--warmup", "2", "--grid", "120", "--submaps", "3",
             "--passes-per-step", "1"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    # The collective backend may print its own banner through C stdio (gloo here, RCCL on the
    # box); bench.py drains that first, so the JSON is the LAST line and the only JSON line.
    lines0 = [l for l in outs[0][0].splitlines() if l.strip()]
    assert [l.startswith("{") for l in lines0].count(True) == 1 and lines0[-1].startswith("{")
    assert not any(l.startswith("{") for l in outs[1][0].splitlines())    # rank 0 only
    out = json.loads(lines0[-1])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["config"]["name"] == "c3" and out["scaling"] == "weak"
    assert out["config"]["submaps_per_gpu"] == 3                # distinct submaps: 6 node-wide
    assert out["config"]["constraints_found_node_wide"] == 6    # all-gathered from both ranks
    assert out["config"]["candidates_per_step"] == 3000.0       # per rank (weak scaling)
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 5 / (2 * 5 * 3000.0) - 1) < 1e-4   # both ranks
    assert "cpu_baseline" not in out                            # N = 1 only
    for _, err in outs:
        assert "issued 8 matches" in err          # warmup + steps + the instrumented step


def test_bench_two_ranks_gloo_c5():
    """`--config c5 --gpus 2` (the 3D loop-closure batch, BASELINE config[4]) on two CPU processes
    over gloo: submaps sharded in equal blocks (rank r owns submaps r * pairs ...), every rank
    searches its block, the optional constraints are all-gathered and the node-wide best match is
    the all-reduce(max) of the packed key; rank 0 prints ONE line."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29551")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(root, "tests", "bench_stub.py"), "--config", "c5",
             "--gpus", "2", "--steps", "4", "--warmup", "1", "--submaps", "3",
             "--passes-per-step", "1"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines0 = [l for l in outs[0][0].splitlines() if l.strip()]
    assert [l.startswith("{") for l in lines0].count(True) == 1 and lines0[-1].startswith("{")
    assert not any(l.startswith("{") for l in outs[1][0].splitlines())    # rank 0 only
    out = json.loads(lines0[-1])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["config"]["name"] == "c5" and out["config"]["submaps_per_gpu"] == 3
    # the stub finds pairs 0 and 2 of every rank's block: 4 constraints node-wide; the best score
    # is the last even pair's (0.52), which both ranks report for their local pair 2 -- the packed
    # key resolves the tie to the LOWEST global submap index: rank 0's
    assert out["config"]["constraints_found_node_wide"] == 4
    assert abs(out["config"]["best_match"]["score"] - 0.52) < 1e-6
    assert out["config"]["best_match"]["submap"] == 2
    assert "cpu_baseline" not in out                            # N = 1 only
    for _, err in outs:
        assert "issued 6 matches" in err


def test_bench_two_ranks_calibrate_together():
    """The driver's own command line (no --passes-per-step) on two ranks of unequal speed: the
    calibrated pass count is agreed on (all-reduce MAX) BEFORE any rank runs a trial step -- a
    sharded pass contains collectives, so ranks running different numbers of passes would
    deadlock -- and both ranks issue exactly the same number of searches."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29549",
                   CMX_STUB_SLEEP_US="1500" if rank == 1 else "100")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(root, "tests", "bench_stub.py"), "--gpus", "2",
             "--steps", "3", "--warmup", "1", "--grid", "120", "--submaps", "2"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    issued = [int(re.search(r"issued (\d+) matches", err).group(1)) for _, err in outs]
    assert issued[0] == issued[1], issued
    out = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    p = out["config"]["passes_per_step"]
    assert p >= 2 and out["n_gpus"] == 2                 # >= 30 ms of the slower rank's passes
    assert out["config"]["candidates_per_step"] == 2000.0 * p
    assert out["ms_per_step"] >= 30.0 * 0.8


def test_bench_calibrates_passes_per_step(stubbed_bench, monkeypatch):
    """Without --passes-per-step a step is sized during warmup (untimed) to last >= 30 ms; the
    stub's searches take microseconds, so many passes make one step, and exactly steps x passes
    timed searches are accounted."""
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--no-other", "--grid",
                                      "120", "--steps", "3", "--warmup", "1", "--concurrency", "2"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    out = json.loads(buf.getvalue().strip().splitlines()[-1])
    p = out["config"]["passes_per_step"]
    assert p > 1 and p % 2 == 0 and out["config"]["host_threads"] == 2
    assert out["config"]["candidates_per_step"] == 1000.0 * p
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 3 / (3 * 1000.0 * p) - 1) < 1e-4
    assert out["config"]["timed_region_s"] > 0          # (stub searches take microseconds)


def test_line_is_small_and_the_details_go_to_a_file(stubbed_bench, tmp_path, monkeypatch):
    """The driver's record keeps a line of a few KB (round 4's 30 KB line came back unparsed):
    the ONE line on stdout is < 4 KB, parses, carries every contract key plus `roofline`,
    `parity` and `summary`; the full record (notes, nested blocks) is in the file it names."""
    import bench
    details = tmp_path / "details.json"
    calls = []
    monkeypatch.setattr(sys, "argv", ["bench.py", "--no-cpu-baseline", "--no-other", "--grid", "120",
                                      "--steps", "3", "--warmup", "1", "--details", str(details)])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, len(lines[0])
    out = json.loads(lines[0])
    for key in REQUIRED + ["parity", "summary", "details"]:
        assert key in out, key
    assert all(not isinstance(v, (dict, list)) or k == "best_match"
               for k, v in out["config"].items())
    full = json.loads(details.read_text())
    assert full["value"] == pytest.approx(out["value"], rel=1e-5)
    assert "note" in full["roofline"] and "note" not in out["roofline"]


def test_headline_of_a_full_record_stays_under_the_limit():
    """A record with every leg of the default run (14 configs, CPU baselines, parity records,
    long workload texts and notes) still makes a line < 4 KB."""
    import bench
    names = ["c1_single", "c1b128", "c1b128_dirty", "c1b128_g400", "c1b1024", "c1b1024_dirty",
             "c1b128t8", "c2_easy", "c3s16", "c3s64", "c4", "c5_single", "c5s32", "c1_tsdf"]
    entry = {"ms": 0.147787290043, "cand_per_s": 3952058393.04, "matches_per_s": 866109.66316,
             "frac": 0.12346093812, "bound": "gather-issue", "cpu": 7270691.4588,
             "cpu_unit": "candidates/s", "cpu_cores": 32, "parity": "exact"}
    out = {"metric": "candidate poses scored/sec", "value": 9107031338.8, "unit": "candidates/s",
           "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 30.80333, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
           "constraints_per_s": 20900.123,
           "config": dict({k: 1234.56789012 for k in bench.CONFIG_KEYS}, workload="w" * 900,
                          name="c2", best_match={"score": 0.71234567, "submap": 137},
                          **{f"c1b128_{i}": 1.0 for i in range(150)}),
           "roofline": dict({k: 0.123456789 for k in bench.ROOFLINE_KEYS}, kernel="k" * 300,
                            bound="l2-gather", unit="GB/s", note="n" * 2000),
           "cpu_baseline": dict({k: 22263134.5335 for k in bench.CPU_KEYS}, unit="candidates/s",
                                kind="reference", sample="s" * 700),
           "parity": {"vs": "reference", "checked": 1, "max_abs_dscore": 0.0, "max_abs_dpose": 0.0,
                      "bit_exact": True, "tol": 1e-4},
           "details": {"x": "y" * 20000},
           "summary": dict({"c2": dict(entry)}, **{n: dict(entry) for n in names})}
    text = bench.headline(out, "gpurun_out/bench_details.json")
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    assert line["cpu_baseline"]["kind"] == "reference" and line["roofline"]["bound"] == "l2-gather"
    assert set(line["summary"]) == set(["c2"] + names)


def test_parity_gate_records_and_aborts():
    """bench.py's gate: equal results make a record (bit_exact, differences 0), a score or pose
    beyond the north star's 1e-4 -- or a found flag that differs -- raises before any timing."""
    import bench
    ok = bench.parity_record("reference", [(1, 0.7886148691177368, [2.15, 7.9, -0.04], True,
                                            0.7886148691177368, [2.15, 7.9, -0.04]),
                                           (0, 0.0, None, False, 0.0, None)])
    assert ok["bit_exact"] and ok["checked"] == 2 and ok["max_abs_dscore"] == 0.0
    near = bench.parity_record("port", [(1, 0.5, [1.0, 2.0, 3.0], True, 0.50001, [1.0, 2.0, 3.00001])])
    assert not near["bit_exact"] and 0 < near["max_abs_dscore"] < 1e-4
    assert bench._parity_word(ok) == "exact" and bench._parity_word(near) != "exact"
    with pytest.raises(bench.ParityError):
        bench.parity_record("reference", [(1, 0.5, [1.0, 2.0, 3.0], True, 0.51, [1.0, 2.0, 3.0])])
    with pytest.raises(bench.ParityError):
        bench.parity_record("reference", [(1, 0.5, [1.0, 2.0, 3.0], True, 0.5, [1.0, 2.1, 3.0])])
    with pytest.raises(bench.ParityError):
        bench.parity_record("reference", [(0, 0.0, None, True, 0.5, [1.0, 2.0, 3.0])])


def test_parity_gate_of_the_headline_workload_against_the_reference(monkeypatch):
    """Fast2DWorkload.parity with the real checker (oracle/_ref where built, else the port): a
    stub device that returns the oracle port's result passes; the same with one pose component
    moved by a cell is stopped."""
    import argparse
    import bench
    from cartographer_amd import scan_matching as sm
    from oracle import pyoracle as orc

    class FakeMatcher:
        def __init__(self, grid, depth, device=0):
            pass
    monkeypatch.setattr(sm, "FastCorrelativeScanMatcher2D", FakeMatcher)
    monkeypatch.setattr(sm, "PointCloudOnDevice", lambda scan, device=0: scan)
    args = argparse.Namespace(submaps=0, grid=120, depth=5, beams=300, min_score=0.5, scans=1)
    w = bench.Fast2DWorkload(args, 0, 0, 1, sharded=False)
    port = orc.FastCorrelativeScanMatcher2D(w.cells0, w.lim0["resolution"], w.lim0["max_x"],
                                            w.lim0["max_y"], 5).match_full_submap(w.scan, 0.5)
    assert port["found"]
    device = (np.array([1], np.int32), np.array([port["score"]], np.float32),
              np.array([port["pose"]], np.float64), {})
    rec = bench.parity_gate(w, device)
    assert rec["bit_exact"] and rec["checked"] == 1 and rec["vs"] in ("reference", "port")
    device[2][0, 0] += 0.05
    with pytest.raises(bench.ParityError):
        bench.parity_gate(w, device)
